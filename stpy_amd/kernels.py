"""
Drop-in for ``stpy.kernels.KernelFunction`` on the Gram-matrix hot path (reference:
stpy/kernels.py:10-261 dispatcher, :136-159 ``kernel``, :368-398 SE, :552-583 ARD,
:811-859 Matern, :917-970 ARD-Matern, :300-320 linear).

Same constructor arguments, same ``kernel(a, b, **kwargs) -> (|b|, |a|)`` orientation, same ``+`` /
``*`` algebra and the same kwargs-override protocol (``kernel(a, b, **{'0': {'gamma': g}})``) that
``Estimator.optimize_params_general`` uses to talk to a kernel (estimator.py:156-171).  The
arithmetic runs in ``stpy_gram`` (stpy_amd/csrc/gram.hip); there is no CPU path.
"""
import math

import torch

from . import _lib

# kernel families implemented on the device; everything else in kernels.py:167-261 is outside
# the hot path and raises.
_SUPPORTED = ("squared_exponential", "ard", "matern", "ard_matern", "linear")
_OUT_OF_SCOPE = ("laplace", "modified_matern", "custom", "tanh", "step", "angsim", "full_covariance_se",
				 "full_covariance_matern", "polynomial", "squared_exponential_per_group", "ard_per_group",
				 "gibbs", "gibbs_custom", "random_map")

_MATERN_KIND = {0.5: _lib.K_MATERN12, 1.5: _lib.K_MATERN32, 2.5: _lib.K_MATERN52}

_const_cache = {}


def _scalar(v):
	"""gamma / kappa may arrive as python numbers or 0-d / 1-element tensors (estimator.py:160-166)."""
	if torch.is_tensor(v):
		return float(v.detach().reshape(-1)[0].item())
	return float(v)


def _dev_const(values, dtype, device, int32=False):
	"""Small constant device arrays (inverse lengthscales, column indices), cached by value."""
	key = (tuple(values), dtype if not int32 else "i32", device.index)
	t = _const_cache.get(key)
	if t is None:
		if len(_const_cache) > 4096:
			_const_cache.clear()
		t = torch.tensor(list(values), dtype=torch.int32 if int32 else dtype, device=device)
		_const_cache[key] = t
	return t


class KernelFunction:

	def __init__(self, kernel_function=None, kernel_name="squared_exponential",
				 freq=None, groups=None, d=1, gamma=1, ard_gamma=None, nu=1.5, kappa=1, map=None, power=2,
				 cov=None, params=None, group=None, offset=0.):
		if kernel_function is not None:
			raise NotImplementedError("custom python kernel functions are outside the stpy_amd hot path")
		self.offset = offset
		self.optkernel = kernel_name
		self.gamma = gamma
		if ard_gamma is None:
			self.ard_gamma = torch.ones(d).double()
		else:
			# kernels.py:36-39: Tensor([ard_gamma]) if that works, else keep as given
			try:
				self.ard_gamma = torch.Tensor([ard_gamma]).double()
			except Exception:
				self.ard_gamma = ard_gamma
		self.power = power
		self.v = nu
		if params is not None:
			self.initial_params = params
		else:
			self.initial_params = {'kappa': kappa}
		self.cov = torch.eye(d).double() if cov is None else cov
		self.group = [i for i in range(d)] if group is None else group
		self.map = map
		self.groups = groups
		self.kappa = kappa
		self.freq = freq
		self.d = d
		self.add = False

		self.params = self._initial_item_params()
		# one entry per kernel item: the KernelFunction object whose attributes are the fallbacks
		# for parameters missing from an override dict (bound-method semantics of kernels.py:68)
		self._owners = [self]
		self.optkernel_list = [self.optkernel]
		self.params_dict = {'0': self.params}
		self.kernel_items = 1
		self.operations = ["-"]

	# ------------------------------------------------------------------ construction helpers
	def _initial_item_params(self):
		"""kernels.py:167-261 (get_kernel_internal): the stored parameter dictionary of one item."""
		params = {**self.initial_params, 'kappa': self.kappa, 'group': self.group, 'offset': self.offset}
		name = self.optkernel
		if name == "squared_exponential":
			params = dict(**params, **{'gamma': self.gamma})
		elif name == "ard" and self.groups is None:
			params = dict(**params, **{'ard_gamma': self.ard_gamma})
		elif name == "linear":
			pass
		elif name == "matern":
			params = dict(**params, **{'gamma': self.gamma, 'nu': self.v})
		elif name == "ard_matern":
			params = dict(**params, **{'ard_gamma': self.ard_gamma, 'nu': self.v})
		elif name in _OUT_OF_SCOPE or (name == "ard" and self.groups is not None):
			raise NotImplementedError("kernel '%s' is outside the stpy_amd hot path (supported: %s)" % (name, ", ".join(_SUPPORTED)))
		else:
			raise AssertionError("Kernel not implemented.")     # kernels.py:261
		return params

	def __combine__(self, second_kernel_object):
		"""kernels.py:76-82."""
		self._owners = self._owners + second_kernel_object._owners
		self.optkernel_list = self.optkernel_list + second_kernel_object.optkernel_list
		self.operations = self.operations + second_kernel_object.operations[1:]
		for key, value in second_kernel_object.params_dict.items():
			self.params_dict[str(self.kernel_items)] = value
			self.kernel_items += 1

	def __add__(self, second_kernel_object):
		"""kernels.py:84-89."""
		self.__combine__(second_kernel_object)
		diff = len(set(second_kernel_object.group) - set(self.group))
		self.d += diff
		self.operations.append("+")
		return self

	def __mul__(self, second_kernel_object):
		"""kernels.py:91-94."""
		self.__combine__(second_kernel_object)
		self.operations.append("*")
		return self

	def description(self):
		"""kernels.py:96-103."""
		desc = "Kernel description:"
		for index in range(0, self.kernel_items, 1):
			desc = desc + "\n\n\tkernel: " + self.optkernel_list[index]
			desc = desc + "\n\toperation: " + self.operations[index]
			desc = desc + "\n\t" + "\n\t".join(
				["{0}={1}".format(key, value) for key, value in self.params_dict[str(index)].items()])
		return desc

	def add_groups(self, dict):
		"""kernels.py:105-110."""
		for a in self.params_dict.keys():
			if a not in dict.keys():
				dict[a] = {}
			dict[a]['group'] = self.params_dict[a]['group']
		return dict

	def get_param_refs(self):
		return self.params_dict

	def get_kernel(self):
		return self.kernel

	# ------------------------------------------------------------------ parameter resolution
	def _resolve(self, kwargs):
		"""
		kernels.py:138-157: with kwargs present they *replace* params_dict (only 'group' is
		re-injected); a key missing from an item's dict falls back to the owning object's attribute.
		Returns one launch description per kernel item.
		"""
		if len(kwargs) > 0:
			params_dict = kwargs
			self.add_groups(params_dict)
		else:
			params_dict = self.params_dict
		items = []
		for i in range(self.kernel_items):
			owner = self._owners[i]
			arg = params_dict[str(i)] if str(i) in params_dict.keys() else {}
			name = self.optkernel_list[i]
			kappa = _scalar(arg['kappa']) if 'kappa' in arg else _scalar(owner.kappa)
			group = list(arg['group']) if 'group' in arg else list(owner.group)
			offset = 0.0
			if name == "squared_exponential":
				gamma = _scalar(arg['gamma']) if 'gamma' in arg else _scalar(owner.gamma)
				kind, inv_ls = _lib.K_SE, [1.0 / gamma] * len(group)
			elif name == "ard":
				g = arg['ard_gamma'] if 'ard_gamma' in arg else owner.ard_gamma
				g = torch.as_tensor(g).detach().double().reshape(-1)
				kind, inv_ls = _lib.K_SE, [1.0 / float(g[j]) for j in group]              # kernels.py:572
			elif name == "matern":
				gamma = _scalar(arg['gamma']) if 'gamma' in arg else _scalar(owner.gamma)
				nu = arg['nu'] if 'nu' in arg else owner.v
				kind, inv_ls = self._matern_kind(nu), [1.0 / gamma] * len(group)
			elif name == "ard_matern":
				g = arg['ard_gamma'] if 'ard_gamma' in arg else owner.ard_gamma
				g = torch.as_tensor(g).detach().double().reshape(-1)
				nu = arg['nu'] if 'nu' in arg else owner.v
				kind, inv_ls = self._matern_kind(nu), [1.0 / float(g[j]) for j in group]  # kernels.py:941
			elif name == "linear":
				offset = _scalar(arg['offset']) if 'offset' in arg else _scalar(owner.offset)
				kind, inv_ls = _lib.K_LINEAR, [1.0] * len(group)
			else:
				raise AssertionError("Kernel not implemented.")
			items.append(dict(kind=kind, kappa=kappa, group=group, inv_ls=inv_ls, offset=offset, op=self.operations[i]))
		return items

	@staticmethod
	def _matern_kind(nu):
		nu = _scalar(nu)
		if nu not in _MATERN_KIND:
			raise NotImplementedError("Matern nu=%s: only 0.5, 1.5, 2.5 run on the device (general-nu Bessel "
									  "form, kernels.py:852-858, is outside the hot path)" % nu)
		return _MATERN_KIND[nu]

	# ------------------------------------------------------------------ evaluation
	def _kernel_into(self, a, b, out, kwargs=None, diag_add=0.0, lower_only=False):
		"""
		Device-side evaluation: a (n, d), b (q, d) and out (q, n) are tensors on this process's
		GPU.  ``diag_add`` (s^2 of gauss_procc.py:151-163) is applied with the last kernel item.
		"""
		lib = _lib.load()
		items = self._resolve(dict(kwargs) if kwargs else {})
		dt = _lib.dtype_code(out.dtype)
		n, q = a.shape[0], b.shape[0]
		dmax = max(len(it['group']) for it in items)
		work = torch.empty((int(lib.stpy_gram_workspace_bytes(dt, n, q, dmax)),), dtype=torch.uint8, device=out.device)
		for idx, it in enumerate(items):
			group = it['group']
			identity = (group == list(range(a.shape[1])))
			cols = None if identity else _dev_const(group, None, out.device, int32=True)
			inv_ls = _dev_const(it['inv_ls'], out.dtype, out.device)
			combine = {"-": _lib.OUT_SET, "+": _lib.OUT_ADD, "*": _lib.OUT_MUL}[it['op']]
			last = idx == len(items) - 1
			rc = lib.stpy_gram(it['kind'], dt, _lib.ptr(a), n, a.stride(0), _lib.ptr(b), q, b.stride(0),
							   len(group), _lib.ptr(cols), _lib.ptr(inv_ls), it['kappa'], it['offset'],
							   diag_add if last else 0.0, 1 if lower_only else 0, combine,
							   _lib.ptr(out), out.stride(0), _lib.ptr(work), _lib.stream_ptr())
			_lib.check(rc, "stpy_gram")
		return out

	def kernel(self, a, b, **kwargs):
		"""kernels.py:136-159.  a: (n, d), b: (q, d)  ->  (q, n); result lives where ``a`` lives."""
		ad = _lib.to_device(a)
		bd = _lib.to_device(b, ad.dtype)
		out = torch.empty((bd.shape[0], ad.shape[0]), dtype=ad.dtype, device=ad.device)
		self._kernel_into(ad, bd, out, kwargs)
		return _lib.like_input(out, a)

	def _diag_into(self, x, out, kwargs=None):
		lib = _lib.load()
		items = self._resolve(dict(kwargs) if kwargs else {})
		dt = _lib.dtype_code(out.dtype)
		for it in items:
			group = it['group']
			identity = (group == list(range(x.shape[1])))
			cols = None if identity else _dev_const(group, None, out.device, int32=True)
			inv_ls = _dev_const(it['inv_ls'], out.dtype, out.device)
			combine = {"-": _lib.OUT_SET, "+": _lib.OUT_ADD, "*": _lib.OUT_MUL}[it['op']]
			rc = lib.stpy_gram_diag(it['kind'], dt, _lib.ptr(x), x.shape[0], x.stride(0), len(group), _lib.ptr(cols),
									_lib.ptr(inv_ls), it['kappa'], it['offset'], combine, _lib.ptr(out), _lib.stream_ptr())
			_lib.check(rc, "stpy_gram_diag")
		return out

	def kernel_self_diag(self, x, **kwargs):
		"""k(x_i, x_i) for every row: what gauss_procc.py:347 assembles with a Python loop; shape (m,)."""
		xd = _lib.to_device(x)
		out = torch.empty((xd.shape[0],), dtype=xd.dtype, device=xd.device)
		self._diag_into(xd, out, kwargs)
		return _lib.like_input(out, x)

	# ------------------------------------------------------------------ finite-dimensional cases (kernels.py:263-273)
	def embed(self, x):
		if self.optkernel == "linear":
			return x
		raise AttributeError("This type of kernel does not support a finite dimensional embedding")

	def get_basis_size(self):
		if self.optkernel == "linear":
			return self.d
		raise AttributeError("This type of kernel does not support a finite dimensional embedding")

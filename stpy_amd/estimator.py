"""
Drop-in for the part of ``stpy.estimator.Estimator`` that CALLS the hot path (reference: stpy/estimator.py:15-40 base class,
load_data, log_marginal; :42-257 ``optimize_params_general``).  ``GaussianProcess`` derives from it as in the reference
(gauss_procc.py:18), and ``GaussianProcess.optimize_params`` builds the ``params`` dictionary and hands it to the generic
driver here exactly as gauss_procc.py:640-702 does.

The driver minimises ``self.log_marginal(self.kernel_object, input_dict, weight)`` -- every evaluation is a full Gram fill +
Cholesky (+ inverse for the gradient) on the device, with the analytic gradient reaching torch autograd through
``_LogMarginalFn`` -- over the variables named in ``params``:

    params = {item_key: {var_name: (init_value | init_func | None, manifold, bounds)}}        (estimator.py:60-68)

``manifold`` only has to expose ``.dim`` (and ``random_point()`` for the steepest-descent branch); ``Euclidean`` below is the
minimal stand-in for ``pymanopt.manifolds.Euclidean`` used when pymanopt is not installed.

Optimisers (estimator.py:70-231):
  "pytorch-minimize"  L-BFGS as in the reference: torchmin when it is importable, otherwise scipy's L-BFGS-B on the same
                      cost / gradient (also what the reference itself switches to when bounds are given, :193-203);
  "pymanopt"          steepest descent with backtracking line search on the product of Euclidean factors: pymanopt's
                      SteepestDescent when importable, otherwise the same iteration written out here;
  "bisection"         one scalar variable on [a, b]: as in the reference (estimator.py:128-138 with
                      optim/custom_optimizers.py:7-75, version 'stop') the root of the COST on [a, b] by 100 halvings -- a
                      if cost(a) < 0, an error where the reference prints "Bisection method fails." and returns None;
  "scipy", "discrete" are not functional in the reference snapshot (undefined names) and raise here.

One deliberate difference: in the reference's "pytorch-minimize" cost the slice counter only advances in the likelihood branch
(estimator.py:160-166), so with more than one variable every kernel parameter -- and the noise -- reads the FIRST slice of
x.  Here every variable reads its own slice, which is what the write-back loop at :237-244 assumes.
"""
import pickle
from abc import ABC, abstractmethod

import numpy as np
import torch


class Euclidean:
	"""Minimal stand-in for ``pymanopt.manifolds.Euclidean(*shape)``: dimension and a standard-normal random point."""

	def __init__(self, *shape):
		self.shape = tuple(int(s) for s in shape) or (1,)
		self.dim = int(np.prod(self.shape))

	def random_point(self):
		return np.random.normal(size=self.shape)


def _manifold_dim(man):
	return int(getattr(man, "dim"))


class Estimator(ABC):

	def fit(self):
		pass

	@abstractmethod
	def ucb(self, x):
		pass

	@abstractmethod
	def lcb(self, x):
		pass

	def load_data(self, d):
		"""estimator.py:28-30."""
		self.x = d[0]
		self.y = d[1]

	# ------------------------------------------------------------------ hyper-parameter search driver
	def optimize_params_general(self, params={}, restarts=2, optimizer="pymanopt", maxiter=1000, mingradnorm=1e-4, regularizer_func=None,
								verbose=False, scale=1., weight=1., save=False, save_name='model.np', parallel=False, cores=None):
		"""estimator.py:42-257.  Returns True after writing the best point back into ``kernel_object.params_dict`` / ``self.s``
		and refitting (``back_prop`` is switched off as in the reference, :250)."""
		slots = []                       # (key, var_name, init, manifold, bound, dim)
		for key, dict_params in params.items():
			for var_name, value in dict_params.items():
				init_value, manifold, bound = value
				slots.append((key, var_name, init_value, manifold, bound, _manifold_dim(manifold)))
		if not slots:
			raise ValueError("optimize_params_general: no variables given")
		dims = np.cumsum([0] + [sl[5] for sl in slots]).astype(int)
		dim = int(dims[-1])

		# The search works on a COPY of the stored parameters (one shallow copy per kernel item): trial points -- slices of a
		# tensor that requires grad -- never land in kernel_object.params_dict, so an exception in the middle of the search (a
		# trial point that is not positive definite, a keyboard interrupt) leaves the object exactly as it was.  Only the noise
		# level has to live on the object while a value is computed (log_marginal reads self.s); it is restored in `finally`.
		trial_dict = {k: dict(v) for k, v in self.kernel_object.params_dict.items()}

		def build_input(xt):
			"""x (flat tensor) -> the override dictionary log_marginal takes; the noise goes to self.s"""
			for c, (key, var, _, _, _, _) in enumerate(slots):
				piece = xt[dims[c]:dims[c + 1]]
				if key != "likelihood":
					trial_dict[key][var] = piece
				else:
					self.s = piece
			return trial_dict

		def cost(xt):
			f = self.log_marginal(self.kernel_object, build_input(xt), weight)
			if regularizer_func is not None:
				f = f + regularizer_func(xt)
			return f

		def fun(xnp):
			xt = torch.tensor(np.asarray(xnp, dtype=np.float64).reshape(-1), dtype=torch.float64, requires_grad=True)
			try:
				f = cost(xt)
			except torch.linalg.LinAlgError:
				# a trial point whose matrix is not positive definite: +inf, zero slope -- every line search backs off from it
				return float("inf"), np.zeros(xt.numel())
			f.backward()
			return float(f.detach().reshape(-1)[0]), xt.grad.detach().numpy().astype(np.float64).reshape(-1)

		def initial_point():
			parts = []
			for (key, var, init, man, bound, k) in slots:
				if init is None:
					parts.append((torch.randn(size=(k, 1)).double().view(-1) ** 2 * scale).numpy())
				elif callable(init):
					parts.append(np.asarray(torch.as_tensor(init(k)).detach().double().reshape(-1).numpy(), dtype=np.float64) * np.ones(k))
				else:
					parts.append(np.asarray(init, dtype=np.float64).reshape(-1) * np.ones(k))
			return np.concatenate(parts)

		s_backup = self.s
		objective_values, objective_params = [], []
		try:
			if optimizer == "pytorch-minimize":
				bounds = slots[0][4]
				try:
					from torchmin import minimize as minimize_torch          # the reference's solver when it is installed
				except Exception:                                            # noqa: BLE001
					minimize_torch = None
				import scipy.optimize
				for rep in range(restarts):
					x_init = initial_point()
					if minimize_torch is not None and bounds is None:
						res = minimize_torch(cost, torch.from_numpy(x_init), method='l-bfgs', tol=1e-10, disp=verbose + 1, options={'max_iter': maxiter, 'gtol': mingradnorm})
						objective_params.append(np.asarray(res.x.detach().numpy(), dtype=np.float64))
						objective_values.append(float(res.fun))
					else:
						res = scipy.optimize.minimize(fun, x_init, jac=True, method='L-BFGS-B', bounds=bounds,
													  options={'maxiter': maxiter, 'gtol': mingradnorm, 'ftol': 1e-12})
						objective_params.append(np.asarray(res.x, dtype=np.float64))
						objective_values.append(float(res.fun))
					if verbose:
						print("restart", rep, "f =", objective_values[-1], "x =", objective_params[-1])
			elif optimizer == "pymanopt":
				for rep in range(restarts):
					x = initial_point() if any(sl[2] is not None for sl in slots) else np.concatenate([np.asarray(sl[3].random_point(), dtype=np.float64).reshape(-1) * scale for sl in slots])
					f, g = fun(x)
					step = 1.0 / max(np.linalg.norm(g), 1e-12)
					for it in range(maxiter):
						gn = np.linalg.norm(g)
						if gn < mingradnorm:
							break
						# backtracking (Armijo) line search along -g, the step grown again after a success
						t = step
						while True:
							xn = x - t * g
							fn, gnew = fun(xn)
							if fn <= f - 1e-4 * t * gn * gn or t < 1e-14:
								break
							t *= 0.5
						if not np.isfinite(fn) or t < 1e-14:
							break
						x, f, g, step = xn, fn, gnew, 2.0 * t
					objective_params.append(x)
					objective_values.append(f)
					if verbose:
						print("restart", rep, "f =", f, "x =", x, "iterations", it)
			elif optimizer == "bisection":
				if dim != 1 or slots[0][4] is None:
					raise ValueError("bisection: exactly one scalar variable with bounds (a, b)")
				a, b = [float(v) for v in slots[0][4]]
				memo = {}

				def g(v):
					if v not in memo:
						memo[v] = float(cost(torch.tensor([v], dtype=torch.float64)).detach().reshape(-1)[0])
					return memo[v]
				if g(a) < 0.:
					root = a
				elif g(a) * g(b) > 0.:
					raise ValueError("Bisection method fails.")          # (the reference prints this and returns None)
				else:
					lo, hi = a, b
					root = None
					for _ in range(100):
						mid = (lo + hi) / 2.
						if g(lo) * g(mid) < 0:
							hi = mid
						elif g(hi) * g(mid) < 0:
							lo = mid
						else:
							root = mid if g(mid) == 0 else lo
							break
					if root is None:
						root = (lo + hi) / 2.
				objective_params.append(np.array([root]))
				objective_values.append(g(root))
			elif optimizer in ("scipy", "discrete"):
				raise NotImplementedError("optimizer='%s' does not run in the reference snapshot either (estimator.py:124-126, :227-229)" % optimizer)
			else:
				raise AssertionError("Optimizer not implemented.")          # estimator.py:231
		finally:
			self.s = s_backup          # whatever happened inside the search, no trial value stays on the object

		if save:
			with open(save_name, 'wb') as fh:
				pickle.dump({'params': objective_params, 'evidence': objective_values, 'repeats': restarts, 'dim': dims, 'param_names': list(params.keys())}, fh)
		best = int(np.argmin(objective_values))
		x_best = torch.from_numpy(np.asarray(objective_params[best], dtype=np.float64))
		self.s = s_backup
		for c, (key, var, _, _, _, _) in enumerate(slots):          # estimator.py:237-244
			if key == "likelihood":
				self.s = x_best[dims[c]:dims[c + 1]]
			else:
				self.kernel_object.params_dict[key][var] = x_best[dims[c]:dims[c + 1]]
		self.optimization_trace = {"values": objective_values, "params": objective_params, "best": best}
		self.back_prop = False                         # estimator.py:250
		self.fitted = False
		if verbose:
			print(self.description())
		self.fit_gp(self.x, self.y)
		return True

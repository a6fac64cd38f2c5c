"""
Row-sharded random-Fourier-feature embed across the GPUs of one node (last row;
reference: stpy/embeddings/embedding.py:225-241 -- every output row depends on its own input row only).

"Replicas only": the frequency table W (m x d: 8 MB at BASELINE config 5) is replicated, the N input rows are split
into contiguous, tile-aligned slabs, one per rank, and every rank embeds its slab with the single-GPU
``RFFEmbedding.embed`` -- no collective touches the data path.  ``gather=True`` (tests, small problems) assembles the
whole matrix on every rank with an all-gather; at config 5 the result is 34 GB and stays sharded.
"""
import torch
import torch.distributed as dist

TILE = 128


def row_range(n, rank, world, align=TILE):
	"""[r0, r1) of the rows rank ``rank`` embeds: contiguous slabs, aligned to the 128-row tile of the embed kernels, sizes
	differing by at most one tile; the last slab takes the ragged tail."""
	tiles = (n + align - 1) // align
	base, extra = divmod(tiles, world)
	t0 = rank * base + min(rank, extra)
	t1 = t0 + base + (1 if rank < extra else 0)
	return min(t0 * align, n), min(t1 * align, n)


class ShardedEmbedding:
	"""Wraps an embedding (``RFFEmbedding``, ``QuadratureEmbedding``, ...: anything with ``embed`` and ``get_m``) whose
	parameters are identical on every rank."""

	def __init__(self, embedding, group=None):
		if not dist.is_initialized():
			raise RuntimeError("ShardedEmbedding needs torch.distributed to be initialised (one process per GPU)")
		self.embedding = embedding
		self.group = group
		self.world = dist.get_world_size(group)
		self.rank = dist.get_rank(group)

	def get_m(self):
		return self.embedding.get_m()

	def local_rows(self, n):
		return row_range(n, self.rank, self.world)

	def embed_local(self, x):
		"""x: the full (n, d) input (replicated) -> (r0, r1, Z[r0:r1]) with Z = embedding.embed(x)."""
		r0, r1 = self.local_rows(x.shape[0])
		return r0, r1, self.embedding.embed(x[r0:r1])

	def embed(self, x, gather=False):
		r0, r1, z = self.embed_local(x)
		if not gather:
			return r0, r1, z
		n, m = x.shape[0], z.shape[1]
		parts = []
		staged = z.is_cuda and dist.get_backend(self.group) == "gloo"
		for r in range(self.world):
			a, b = row_range(n, r, self.world)
			buf = z.contiguous() if r == self.rank else torch.empty((b - a, m), dtype=z.dtype, device="cpu" if staged else z.device)
			if staged:
				buf = buf.cpu()
			dist.broadcast(buf, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
			parts.append(buf.to(z.device))
		return torch.cat(parts)

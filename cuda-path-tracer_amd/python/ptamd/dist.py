"""Multi-GPU rendering: one process per GPU, scene replicated, the sample range of every pixel
sharded over ranks (or, for renders of few samples, the image's tiles: shard_work), per-rank accumulation buffers combined by one sum-reduce (RCCL over xGMI via
torch.distributed's "nccl" backend; "gloo" on CPU for tests).

The reference is single-GPU; its own decomposition gives n_blocks_per_tile x n_split_buffers = 32
sample slices per pixel (src/renderer.cu:240-241), which is the axis sharded here: rank r renders
the contiguous sample range shard_samples(spp, r, world).  Because sample s of pixel p always uses
the Philox stream (seed; p, s), the union over ranks is exactly the single-GPU sample set and the
reduced image differs from the single-GPU one only by floating-point summation order.
"""
import numpy as np


def shard_samples(n_samples, rank, world_size, sample_begin=0):
    """Contiguous, balanced shard [begin, end) of the sample range for `rank`."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    b = sample_begin + (n_samples * rank) // world_size
    e = sample_begin + (n_samples * (rank + 1)) // world_size
    return b, e


MIN_SAMPLES_PER_RANK = 8     # one sample slice of the render kernel's work items


def shard_work(n_samples, rank, world_size, sample_begin=0):
    """What `rank` renders: ((sample_begin, sample_end), (tile_rank, tile_ranks)).  With at least MIN_SAMPLES_PER_RANK
    samples per rank the sample range is cut (every rank traces every pixel: identical cache behaviour, perfect balance);
    with fewer, every rank renders ALL samples of its interleaved share of the 8x8-pixel tiles -- the tile axis of the
    reference's grid (src/renderer.cu:219-223) -- so that a rank's launch is not shorter than the kernel's start-up."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    if n_samples >= MIN_SAMPLES_PER_RANK * world_size:
        return shard_samples(n_samples, rank, world_size, sample_begin), (0, 1)
    return (sample_begin, sample_begin + n_samples), (rank, world_size)


def effective_spp(spp, n_split_buffers=4, n_blocks_per_tile=8):
    """Renderer::set_samples_per_pixel + the normalisation in Renderer::render
    (src/renderer.cu:181-183, 270-271): spp is rounded down to a multiple of 32, minimum 32."""
    per_pass = max(spp // (n_blocks_per_tile * n_split_buffers), 1)
    return n_split_buffers * n_blocks_per_tile * per_pass


def reduce_image(local_sum, dist=None, dst=0):
    """Sum-reduces the per-rank radiance sums (torch tensor, any device) onto rank `dst`.
    Without an initialised process group this is the identity."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_sum
    dist.reduce(local_sum, dst=dst, op=dist.ReduceOp.SUM)
    return local_sum


def render_sharded(render_fn, n_samples, rank, world_size, dist=None, dst=0):
    """render_fn(sample_begin, sample_end) -> torch tensor with this rank's un-normalised sum.
    Returns the reduced sum (valid on rank dst) and this rank's sample range."""
    b, e = shard_samples(n_samples, rank, world_size)
    local = render_fn(b, e)
    return reduce_image(local, dist, dst), (b, e)

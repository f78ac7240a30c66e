"""Image-tile sharding across the GPUs of one node (no reference counterpart:
the reference is single-process).  One process per GPU; rank r renders the 32x32 pixel tiles t
with t % world == r into a full-size zeroed raw film (global lane ids, so the image does not
depend on the partition), ONE RCCL all-reduce (sum, f32, H*W*C) merges the films over xGMI, and
the film is developed (divide by W) after the reduction because develop is not linear in the
partial sums.  PRB gradients are reduced the same way (7 floats)."""
import numpy as np

TILE = 32


def tile_pixels(rank, world, width, height):
    """Row-major pixel indices owned by `rank` (mirror of ensure_pixel_list() in csrc/device.hip)."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        ys = np.arange(y0, min(y0 + TILE, height))[:, None]; xs = np.arange(x0, min(x0 + TILE, width))[None, :]
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def develop(raw):
    """HDRFilm::develop (src/films/hdrfilm.cpp:306-410) on a HOST raw film (numpy / CPU tensor): RGB[A] / W, W == 0 -> 1.  Used
    only by the CPU rehearsal of render_distributed (a `render_rank_fn` on gloo, no GPU in the process); on a GPU the reduced film
    is developed by the library's kernel (Scene.develop -> lrt_film_develop -> k_develop)."""
    w = raw[..., -1:]
    w = w + (w == 0)
    return raw[..., :-1] / w


def render_distributed(scene, render_rank_fn=None, spp=0, seed=0, group=None, **kw):
    """Render `scene` across the ranks of the default (or given) process group.

    render_rank_fn(rank, world, film) fills the rank's partial raw film (a torch tensor); the
    default calls the HIP back-end on the rank's GPU.  Returns (image, raw_film) torch tensors,
    identical on every rank."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    h, w, _ = scene.film_shape()
    C = scene.raw_channels()
    if render_rank_fn is None:
        dev = torch.device("cuda", torch.cuda.current_device())
        film = torch.zeros((h, w, C), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        scene.render_to_device(film.data_ptr(), None, spp=spp, seed=seed, tile_rank=rank, tile_count=world,
                               device=dev.index, **kw)
    else:
        film = torch.zeros((h, w, C), dtype=torch.float32)
        render_rank_fn(rank, world, film)
    if world > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    if film.is_cuda:                                  # develop after the reduction, on the device, through the C ABI
        image = torch.empty((h, w, C - 1), dtype=torch.float32, device=film.device)
        torch.cuda.synchronize(film.device)           # the all-reduce ran on PyTorch's stream, the library has its own
        scene.develop(film_ptr=film.data_ptr(), image_ptr=image.data_ptr())
        return image, film
    return develop(film), film


def reduce_gradients(grads, group=None):
    """Sum the per-rank PRB parameter gradients (d sigma_t[3], d albedo[3], d g)."""
    import torch
    import torch.distributed as dist
    v = torch.tensor(np.concatenate([grads["sigma_t"], grads["albedo"], [grads["g"]]]).astype(np.float32))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            v = v.cuda()
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    v = v.cpu().numpy()
    return {"sigma_t": v[:3], "albedo": v[3:6], "g": float(v[6])}

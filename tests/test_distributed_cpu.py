"""The N>1 path on the CPU: world_size-2 gloo processes, each rendering its row-interleaved share
(with the oracle, the only renderer that exists without a GPU), gathered on rank 0 and assembled.
Covers the partition arithmetic, the frame descriptors and the gather ordering that bench.py and
ShardedRenderer use on the GPU."""
import os
import sys

import numpy as np
import pytest

from raytracing_amd.distributed import assemble_rows, owned_rows, rank_frame, rows_per_rank

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_arithmetic():
    for h in (1, 2, 7, 600, 2160, 4320):
        for g in (1, 2, 3, 4, 8):
            rows = [owned_rows(h, r, g) for r in range(g)]
            assert sorted(sum(rows, [])) == list(range(h))        # a partition
            assert max(map(len, rows)) == rows_per_rank(h, g)      # buffers sized for the largest share
            assert max(map(len, rows)) - min(map(len, rows)) <= 1  # balanced
            for r in range(g):
                f = rank_frame(10, h, 4, r, g)
                assert (f.row_first, f.row_step) == (r, g)


def test_frame_rows_matches_partition(oracle_backend):
    for h in (1, 5, 600):
        for g in (1, 2, 4, 8):
            for r in range(g):
                ctx_rows = oracle_backend.lib.rtx_frame_rows(rank_frame(16, h, 4, r, g))
                assert ctx_rows == len(owned_rows(h, r, g))


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from conftest import make_default_context
    from raytracing_amd.rtx import ASSETS_DIR, Backend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = Backend(oracle.ORACLE_LIB_PATH)
    tex = {"water": b.decode_jpeg(os.path.join(ASSETS_DIR, "water.jpg")), "map.jpg": b.decode_jpeg(os.path.join(ASSETS_DIR, "map.jpg"))}
    ctx = make_default_context(b, tex, mask=17, sky="map.jpg")
    W, H, D = 96, 71, 2  # odd height: ranks own different row counts
    f = rank_frame(W, H, D, rank, world)
    mine = ctx.render(b.default_camera(), W, H, D, f.row_first, f.row_step)
    rpr = rows_per_rank(H, world)
    local = torch.zeros(rpr, W, 3, dtype=torch.uint8)
    local[: mine.shape[0]] = torch.from_numpy(mine)
    gathered = [torch.zeros_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, gather_list=gathered, dst=0)
    if rank == 0:
        frame = assemble_rows([g.numpy() for g in gathered], H)
        whole = ctx.render(b.default_camera(), W, H, D)
        np.save(out_path, np.stack([frame, whole]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_interleaved_gather_reassembles_the_frame(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frames.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    frame, whole = np.load(out)
    assert np.array_equal(frame, whole)


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_group_api_reassembles_the_frame_on_the_oracle(oracle_backend, textures, n_ranks):
    """rtx_group_* on the oracle build (a "device" is just another context): rank r renders rows r, r + n, ...;
    the group's frame equals the frame of a single context, and the ray counts add up."""
    from conftest import make_default_context
    from raytracing_amd.rtx import OPT_STATS, Group
    W, H, D = 80, 53, 3
    cam = oracle_backend.default_camera()
    single = make_default_context(oracle_backend, textures, mask=17, sky="map.jpg", options=[(OPT_STATS, 1)])
    want, want_st = single.render(cam, W, H, D, want_stats=True)
    g = Group(oracle_backend, [0] * n_ranks)
    for ctx in g.contexts:
        ctx.load_default_scene(textures["water"], textures["map.jpg"], mask=17)
        ctx.set_option(OPT_STATS, 1)
    g.commit()
    got, st = g.render(cam, W, H, D, want_stats=True)
    assert np.array_equal(got, want)
    assert (st.primary_rays, st.secondary_rays, st.shadow_rays) == (want_st.primary_rays, want_st.secondary_rays, want_st.shadow_rays)
    g.close()

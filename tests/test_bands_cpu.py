"""Multi-rank path on CPU: world_size-2 (and 3) gloo processes each produce their row-interleaved
bands of one frame, all-gather them with equal padded counts exactly as bench.py does over RCCL,
and de-interleave; the result must equal the single-rank frame byte for byte.  The CPU oracle
stands in for the kernel (tests may use it); the partition/gather/de-interleave code is the
product's (ray-tracer_amd/python/bands.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bands

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, height, width, out_dir):
    for p in (os.path.join(HERE, "..", "ray-tracer_amd", "python"), os.path.join(HERE, "..", "oracle"),
              os.path.join(HERE, "..", "tools"), HERE):
        sys.path.insert(0, os.path.abspath(p))
    import bands as b
    import ora
    import parity
    import yaml_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = parity.scene_text("S1", width, height)
    w, cams = yaml_scene.load(text, ora.API)
    rows = b.rank_rows(height, rank, world)
    xs = np.tile(np.arange(width), len(rows)); ys = np.repeat(rows, width)
    img = ora.render(w, cams[0], 5, pixels=(xs, ys))["rgb"].reshape(len(rows), width, 3)
    pad = b.rows_padded(height, world)
    local = torch.zeros((pad, width, 3), dtype=torch.uint8); local[:len(rows)] = torch.from_numpy(img)
    gathered = torch.zeros((world, pad, width, 3), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered.view(-1), local.view(-1))
    full = b.deinterleave(gathered, height)
    # the C ABI's host-side de-interleave (rtx_band_deinterleave / rtx_band_rows_padded: the row mapping the device
    # kernel of rtx_render_gathered uses) must give the same frame
    import rtx
    assert rtx.LIB.rtx_band_rows_padded(height, world) == pad
    assert np.array_equal(rtx.band_deinterleave(gathered.numpy(), world, width, height), full.numpy())
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_banded_frame_equals_single_rank(tmp_path, world):
    import ora
    import parity
    import yaml_scene
    height, width = 52, 64            # 7 bands, the last one partial
    port = 29531 + world
    mp.spawn(_worker, args=(world, port, height, width, str(tmp_path)), nprocs=world, join=True)
    w, cams = yaml_scene.load(parity.scene_text("S1", width, height), ora.API)
    ref = ora.render(w, cams[0], 5)["rgb"]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, ref)


def test_c_abi_deinterleave_matches_python():
    import rtx
    rng = np.random.default_rng(7)
    for h, w, n, bpp in ((52, 64, 2, 3), (100, 33, 3, 3), (8, 5, 4, 3), (2160, 16, 8, 4), (17, 9, 1, 3)):
        pad = bands.rows_padded(h, n)
        assert rtx.LIB.rtx_band_rows_padded(h, n) == pad
        g = rng.integers(0, 256, (n, pad, w, bpp), dtype=np.uint8)
        assert np.array_equal(rtx.band_deinterleave(g, n, w, h, bpp), bands.deinterleave(g, h))


def test_band_bookkeeping():
    for h in (1, 8, 9, 100, 2160):
        for n in (1, 2, 3, 4, 8):
            rows = np.concatenate([bands.rank_rows(h, r, n) for r in range(n)])
            assert sorted(rows.tolist()) == list(range(h))
            assert max(len(bands.rank_rows(h, r, n)) for r in range(n)) <= bands.rows_padded(h, n)

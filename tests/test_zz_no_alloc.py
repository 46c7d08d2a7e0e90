"""rtx_render_device_noalloc (include/rtx.h): the entry point for callers that cannot allocate.  Its ray counters
and tile-order lists live in the library's file-scope device array (one copy per GPU, address from
hipGetSymbolAddress at rtx_init) instead of hipMalloc'ed buffers.  It must write exactly what rtx_render_device
writes: compared bit for bit (RGB8 and hit ids) through the C ABI on two calls in a row, the second with a smaller
input, with the tile-order probe and the split tiles forced on so that every part of the scratch array is in use;
sizes beyond the stated maximum are rejected with RTX_E_CAPACITY."""
import ctypes as C

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu


class _Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.lib.hipStreamDestroy.argtypes = [C.c_void_p]

    def malloc(self, n, fill):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), n) == 0
        assert self.lib.hipMemset(p, fill, n) == 0
        return p

    def to_host(self, p, n, dtype):
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, n, 2) == 0      # hipMemcpyDeviceToHost
        return out


def test_noalloc_entry_point_matches_render_device():
    import rtx
    hip = _Hip()
    ctx = rtx.Context(0)
    st = C.c_void_p()
    assert hip.lib.hipStreamCreate(C.byref(st)) == 0
    bufs = []
    try:
        # first call: a frame large enough for the tile lists (probe launch + class lists + split tiles, all forced on);
        # second call: a smaller, ragged frame of another scene (CSG kernel), bands 1 of 3, compact
        cases = [("S4", 640, 360, (0, 1), 0, dict(lpt=1, split=1, split_class=4)),
                 ("S5", 333, 171, (1, 3), rtx.F_BANDS_COMPACT, dict(lpt=1, split=1, split_class=2)),
                 ("S1", 101, 53, (0, 1), 0, dict(lpt=-1, split=-1, split_class=7))]
        for name, w, h, bands, flags, opts in cases:
            pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
            sc = ctx.scene(pw)
            try:
                ctx.options(**opts)
                rows = rtx.LIB.rtx_band_rows(h, rtx.Bands(*bands)) if flags & rtx.F_BANDS_COMPACT else h
                want_rgb = hip.malloc(rows * w * 3, 0x11); want_hit = hip.malloc(rows * w * 4, 0x22)
                got_rgb = hip.malloc(rows * w * 3, 0x11); got_hit = hip.malloc(rows * w * 4, 0x22)
                bufs += [want_rgb, want_hit, got_rgb, got_hit]
                stats = sc.render_device(pc[0], want_rgb.value, want_hit.value, 5, bands=bands, flags=flags, stream=st.value, stats=True)
                assert stats.rays > 0
                sc.render_device_noalloc(pc[0], got_rgb.value, got_hit.value, 5, bands=bands, flags=flags, stream=st.value)
                ctx.poll(wait=True)                      # waits for the queued render; raises if it overflowed
                a = hip.to_host(want_rgb, rows * w * 3, np.uint8); b = hip.to_host(got_rgb, rows * w * 3, np.uint8)
                assert np.array_equal(a, b), (name, "rgb")
                assert (a != 0x11).any()
                a = hip.to_host(want_hit, rows * w * 4, np.int32); b = hip.to_host(got_hit, rows * w * 4, np.int32)
                assert np.array_equal(a, b), (name, "hit ids")
            finally:
                sc.close()
        # beyond the stated maximum: an 8K frame is 518 400 tiles > RTX_NOALLOC_MAX_TILES
        pw, pc = rtx.load_yaml(parity.scene_text("S1", 7680, 4320))
        sc = ctx.scene(pw)
        try:
            with pytest.raises(rtx.RtxError) as e:
                sc.render_device_noalloc(pc[0], bufs[0].value, 0, 5)
            assert e.value.code == rtx.E_CAPACITY and "RTX_NOALLOC_MAX_TILES" in str(e.value)
        finally:
            sc.close()
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7)
        for b in bufs:
            hip.lib.hipFree(b)
        hip.lib.hipStreamDestroy(st)
        ctx.close()

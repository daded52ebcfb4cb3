"""The C-ABI library loads and exports every symbol include/i2v_hip.h declares; the ctypes structures match the
C structs byte for byte (checked by compiling the header with gcc).  No kernel is launched."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "i2v_hip.h")


@pytest.fixture(scope="module")
def lib():
    import i2v_adapter_unofficial_amd as pkg
    if not os.path.exists(pkg._lib.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return pkg._lib


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(i2v_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    handle = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/i2v_hip.h but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(lib.SIGNATURES) == set(syms)
    assert handle.i2v_abi_version() == lib.ABI_VERSION


def test_ctypes_structs_match_header(lib):
    names = {"i2v_gemm_params": lib.GemmParams, "i2v_attn_params": lib.AttnParams, "i2v_tattn_params": lib.TAttnParams,
             "i2v_gn_params": lib.GnParams, "i2v_ln_params": lib.LnParams, "i2v_motion_attn_params": lib.MotionAttnParams,
             "i2v_cross_attn_fused_params": lib.CrossAttnFusedParams, "i2v_ff_fused_params": lib.FfFusedParams}
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in names.items():
        prog.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            prog.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(prog))
        subprocess.run(["gcc", "-std=c99", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, cls in names.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_bad_arguments_return_error_codes_without_a_gpu(lib):
    """argument validation happens on the host before any launch: callable with no GPU."""
    h = lib.load()
    assert h.i2v_gemm_f16(None, None) == -1
    assert b"null params" in h.i2v_last_error()
    p = lib.GemmParams()
    assert h.i2v_gemm_f16(C.byref(p), None) == -1
    assert h.i2v_attention_f16(None, None) == -1 and h.i2v_layernorm_f16(None, None) == -1
    # 2 images x 300 pixels: 16-row chunks (19 of them) of per-channel (mean, M2) + per-(image, channel) (scale, shift) +
    # per-(image, chunk, group <= 64) (mean, M2)
    assert h.i2v_groupnorm_workspace_bytes(2, 300, 64) == (2 * 19 * 64 * 2 + 2 * 64 * 2 + 2 * 19 * 64 * 2) * 4
    # the fused motion-module attention: which shapes it takes, how many packed weight rows, and the refusal of the others
    assert h.i2v_motion_attn_supported(131072, 320, 8, 40, 16) == 1 and h.i2v_motion_attn_supported(131072 + 16, 320, 8, 40, 16) == 0
    assert h.i2v_motion_attn_supported(32768, 640, 8, 80, 16) == 0 and h.i2v_motion_attn_supported(65536, 320, 8, 40, 4) == 0
    assert h.i2v_motion_attn_pack_rows(8, 40) == 8 * 3 * 48
    assert h.i2v_motion_attn_f16(None, None) == -1
    mp = lib.MotionAttnParams()
    mp.rows, mp.channels, mp.heads, mp.head_dim, mp.frames = 256, 640, 8, 80, 16
    assert h.i2v_motion_attn_f16(C.byref(mp), None) == -1 and b"null pointer" in h.i2v_last_error()
    assert h.i2v_cross_attn_fused_supported(131072, 320, 8, 40, 77, 65536) == 1 and h.i2v_cross_attn_fused_pack_rows(8, 40) == 8 * 48
    assert h.i2v_cross_attn_fused_supported(131072, 320, 8, 40, 81, 65536) == 0 and h.i2v_cross_attn_fused_f16(None, None) == -1
    # (r5) 8 and 32 frames too: two pixels per MFMA tile / two tiles per pixel
    assert h.i2v_motion_attn_supported(65536, 320, 8, 40, 8) == 1 and h.i2v_motion_attn_supported(65536, 320, 8, 40, 32) == 1
    assert h.i2v_ff_fused_supported(131072, 320, 1280) == 1 and h.i2v_ff_fused_supported(32768, 640, 2560) == 0
    # (ABI 8) the block's closing Linear as the tail of the fused feed-forward: rows in (batch, pixel, frame) order need whole clips
    # of a power-of-two frame count
    assert h.i2v_ff_fused_tail_supported(131072, 320, 1280, 0, 0) == 1 and h.i2v_ff_fused_tail_supported(131072, 320, 1280, 16, 4096) == 1
    assert h.i2v_ff_fused_tail_supported(131072, 320, 1280, 12, 4096) == 0 and h.i2v_ff_fused_tail_supported(131072, 320, 1280, 16, 4095) == 0
    fp = lib.FfFusedParams()
    fp.rows, fp.channels, fp.inner = 256, 320, 1280
    assert h.i2v_ff_fused_f16(C.byref(fp), None) == -1 and b"null pointer" in h.i2v_last_error()
    # (ABI 8) LayerNorm 1 + q | k | q_adapter + V^T in one launch
    assert h.i2v_ln_qkv_supported(131072, 320, 960, 4096) == 1 and h.i2v_ln_qkv_supported(131072, 320, 640, 4096) == 1
    assert h.i2v_ln_qkv_supported(8192, 320, 320, 4096) == 1 and h.i2v_ln_qkv_supported(32768, 640, 1920, 1024) == 0      # (n_qk = C: the adapter's K0 | V0^T)
    assert h.i2v_ln_qkv_supported(131072, 320, 1280, 4096) == 0
    assert h.i2v_ln_qkv_supported(131072, 320, 960, 4000) == 0 and h.i2v_ln_qkv_f16(None, None) == -1
    qp = lib.LnQkvParams()
    qp.rows, qp.channels, qp.n_qk, qp.rows_per_image = 256, 320, 960, 128
    assert h.i2v_ln_qkv_f16(C.byref(qp), None) == -1 and b"null pointer" in h.i2v_last_error()
    assert h.i2v_ff_fused_f16(None, None) == -1
    assert h.i2v_colsum_workspace_bytes(1000, 70) == 4 * 70 * 4 and h.i2v_colsum_det_f32(None, 0, None, 0, None, 0, 0, None, None) == -1


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libi2v_hip.so")
    with pytest.raises(lib.HipLibraryError, match="no CPU fallback"):
        lib.load()


def test_model_handle_registry_and_plan_without_a_gpu(lib):
    """the model handle of SURVEY 8(b) (include/i2v_hip.h "Model handle"): create / set_weight / get_weight / plan / destroy and
    their error codes work on the host alone; capture and replay need a stream (tests/test_kernels_gpu.py)."""
    h = lib.load()
    cfg = lib.UnetConfig(4, 4, (C.c_int32 * 4)(320, 640, 1280, 1280), 2, 8, 768, 32, 32, 8, 1, 0)
    hd = C.c_void_p()
    assert h.i2v_unet_create(C.byref(cfg), C.byref(hd)) == 0 and hd.value
    assert h.i2v_unet_create(None, C.byref(hd)) == -1 and b"null argument" in h.i2v_last_error()
    bad = lib.UnetConfig(4, 4, (C.c_int32 * 4)(320, 650, 1280, 1280), 2, 8, 768, 32, 32, 8, 1, 0)
    other = C.c_void_p()
    assert h.i2v_unet_create(C.byref(bad), C.byref(other)) == -1 and b"block_out_channels[1]" in h.i2v_last_error()
    buf = (C.c_uint16 * (320 * 4 * 9))()
    shape = (C.c_int64 * 4)(320, 4, 3, 3)
    assert h.i2v_unet_set_weight(hd, b"conv_in.weight", C.addressof(buf), 0, 4, shape) == 0
    assert h.i2v_unet_set_weight(hd, b"conv_in.weight", C.addressof(buf), 7, 4, shape) == -1 and b"dtype" in h.i2v_last_error()
    assert h.i2v_unet_set_weight(hd, b"", C.addressof(buf), 0, 4, shape) == -1
    assert h.i2v_unet_num_weights(hd) == 1
    p, dt, nd, shp = C.c_void_p(), C.c_int32(-1), C.c_int32(-1), (C.c_int64 * 4)()
    assert h.i2v_unet_get_weight(hd, b"conv_in.weight", C.byref(p), C.byref(dt), C.byref(nd), shp) == 0
    assert p.value == C.addressof(buf) and dt.value == 0 and nd.value == 4 and list(shp) == [320, 4, 3, 3]
    assert h.i2v_unet_get_weight(hd, b"nope", C.byref(p), None, None, None) == 0 and p.value is None
    # the plan: frames within the positional table (unet:725), latent sizes that halve exactly three times (pipe:213-214)
    assert h.i2v_unet_activation_bytes(hd) == 0 and h.i2v_unet_has_step(hd) == 0
    assert h.i2v_unet_set_plan(hd, b"x" * 200, 200) == -1 and b"i2v_unet_plan" in h.i2v_last_error()      # the problem first
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 64, 64, 77, 0))) == 0
    assert h.i2v_unet_activation_bytes(hd) == 0                    # ... is the installed launch plan's arena: none yet
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == -1 and b"no launch plan" in h.i2v_last_error()
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 33, 64, 64, 77, 0))) == -1 and b"positional table" in h.i2v_last_error()
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 60, 63, 77, 0))) == 0      # forward_upsample_size sizes are the plan's business
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 0, 64, 77, 0))) == -1 and b"non-positive" in h.i2v_last_error()
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 64, 64, 81, 1))) == -1 and b"image tokens" in h.i2v_last_error()
    assert h.i2v_unet_replay_step(hd, None) == -1 and b"no captured step" in h.i2v_last_error()
    assert h.i2v_unet_end_capture(hd) == -1 and h.i2v_unet_capture_step(hd, None) == -1
    assert h.i2v_unet_destroy(hd) == 0 and h.i2v_unet_destroy(None) == 0


def _plan_blob(lib, ops, keys=(), arena=0, problem=(2, 16, 64, 64, 77, 0), abi=None, io_mask=0):
    """a launch plan by hand THROUGH THE RECORDER'S OWN WRITER (handle.pack_op / pack_plan: the format `record_plan` emits, parsed by
    csrc/handle.hip): ops = [(entry id, struct bytes, [slot integers], [(payload offset, kind, index, addend)])]"""
    import struct
    from i2v_adapter_unofficial_amd import handle as H
    ops_bin, relocs, payload = [], [], bytearray()
    for entry, sbytes, slots, rel in ops:
        ops_bin.append(H.pack_op(entry, sbytes, [struct.pack("<q", v) for v in slots], rel, payload, relocs))
    blob = bytearray(H.pack_plan(list(keys), ops_bin, relocs, payload, problem, arena, io_mask))
    if abi is not None:
        blob[8:12] = struct.pack("<I", abi)
    return bytes(blob)


def test_model_handle_launch_plan_without_a_gpu(lib):
    """i2v_unet_set_plan / i2v_unet_forward (ABI 9; unet:1289-1451 as a launch plan): what the library checks before anything is
    launched -- the blob's magic / ABI / problem / struct sizes / relocation targets, unregistered weight keys, missing arguments and
    arena -- on hand-made plans; a launch whose own argument check fails returns that entry point's error.  No kernel runs here."""
    import struct
    from i2v_adapter_unofficial_amd import handle as H
    assert list(H.ENTRY_IDS.values()) == list(range(len(H.ENTRY_IDS)))
    h = lib.load()
    cfg = lib.UnetConfig(4, 4, (C.c_int32 * 4)(320, 640, 1280, 1280), 2, 8, 768, 32, 32, 8, 1, 0)
    hd = C.c_void_p()
    assert h.i2v_unet_create(C.byref(cfg), C.byref(hd)) == 0
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 64, 64, 77, 0))) == 0
    sp = lambda b: h.i2v_unet_set_plan(hd, b, len(b))
    empty = _plan_blob(lib, [])
    assert sp(empty) == 0 and h.i2v_unet_plan_launches(hd) == 0 and h.i2v_unet_plan_num_keys(hd) == 0
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == 0          # nothing to launch, nothing read
    assert sp(empty[:50]) == -1 and b"not a launch plan" in h.i2v_last_error()
    assert sp(b"XXXX" + empty[4:]) == -1 and b"magic" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [], abi=lib.ABI_VERSION - 1)) == -1 and b"recorded against ABI" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [], problem=(2, 8, 64, 64, 77, 0))) == -1 and b"recorded for" in h.i2v_last_error()
    assert sp(empty + b"\0" * 8) == -1 and b"bytes" in h.i2v_last_error()
    silu = H.ENTRY_IDS["i2v_silu_f16"]
    assert sp(_plan_blob(lib, [(99, b"", [0, 0, 0], [])])) == -1 and b"entry point 99" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [(silu, b"", [0, 0], [])])) == -1 and b"another header" in h.i2v_last_error()      # a slot short
    gemm = H.ENTRY_IDS["i2v_gemm_f16"]
    assert sp(_plan_blob(lib, [(gemm, b"\0" * (C.sizeof(lib.GemmParams) - 8), [], [])])) == -1 and b"another header" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [(silu, b"", [0, 0, 8], [(24, 0, 0, 0)])])) == -1 and b"relocation" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [(silu, b"", [0, 0, 8], [(0, 1, 3, 0)])], keys=(b"a#w",))) == -1 and b"names key 3" in h.i2v_last_error()
    assert sp(_plan_blob(lib, [(silu, b"", [0, 0, 8], [(0, 0, 0, 4096)])], arena=1024)) == -1 and b"outside the arena" in h.i2v_last_error()
    # a plan of one launch: silu(weight `blk#w` -> arena + 256), 8 elements
    one = _plan_blob(lib, [(silu, b"", [0, 0, 8], [(0, 1, 0, 16), (8, 0, 0, 256)])], keys=(b"blk#w",), arena=1024, io_mask=0)
    assert sp(one) == 0 and h.i2v_unet_plan_launches(hd) == 1 and h.i2v_unet_plan_key(hd, 0) == b"blk#w" and h.i2v_unet_plan_key(hd, 1) is None
    assert h.i2v_unet_activation_bytes(hd) == 1024
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == -1 and b"arena" in h.i2v_last_error()
    assert h.i2v_unet_set_workspace(hd, 4096 + 8, 2048) == -1 and b"256-byte aligned" in h.i2v_last_error()
    assert h.i2v_unet_set_workspace(hd, 4096, 512) == 0
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == -1 and b"the plan needs 1024" in h.i2v_last_error()
    assert h.i2v_unet_set_workspace(hd, 4096, 2048) == 0
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == -1 and b"`blk#w` of the launch plan is not registered" in h.i2v_last_error()
    # an argument the plan reads must be passed
    needs_ctx = _plan_blob(lib, [(silu, b"", [0, 0, 8], [(0, 2, 2, 0), (8, 0, 0, 0)])], arena=1024, io_mask=1 << 2)
    assert sp(needs_ctx) == 0
    assert h.i2v_unet_forward(hd, None, None, None, None, None, None) == -1 and b"`context` (argument 2) is NULL" in h.i2v_last_error()
    # the general entry: i2v_unet_run with the arguments as an array (i2v_unet_forward is it with the forward's five)
    arr = (C.c_void_p * 5)(None, None, None, None, None)
    assert h.i2v_unet_run(hd, arr, 5, None) == -1 and b"argument 2" in h.i2v_last_error()
    assert h.i2v_unet_run(hd, arr, 9, None) == -1 and h.i2v_unet_run(hd, None, 3, None) == -1
    # a new problem drops the plan; the same problem keeps it
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 16, 64, 64, 77, 0))) == 0 and h.i2v_unet_plan_launches(hd) == 1
    assert h.i2v_unet_plan(hd, C.byref(lib.UnetPlan(2, 8, 64, 64, 77, 0))) == 0 and h.i2v_unet_plan_launches(hd) == 0
    assert h.i2v_unet_abort_capture(hd) == 0 and h.i2v_unet_abort_capture(None) == -1
    assert h.i2v_unet_destroy(hd) == 0

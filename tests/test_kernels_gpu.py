"""GPU parity of every C-ABI kernel against plain fp32 torch on the CPU (same fp16-rounded inputs).

Tolerance (stated, fp16 path): outputs are fp16 (10-bit mantissa, eps = 9.8e-4) with fp32 accumulation, so each
kernel must match the fp32 reference within  |err| <= 3e-3 * |ref| + 3e-3 * max|ref|  (elementwise kernels:
2e-3).  Everything is called through the C ABI (i2v_adapter_unofficial_amd.kernels -> ctypes -> libi2v_hip.so).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def K():
    import i2v_adapter_unofficial_amd as pkg
    return pkg.kernels


def h(t):
    """fp16-rounded fp32 copy (what the kernel actually sees)."""
    return t.half().float()


def close(got, ref, rel=3e-3, name=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    tol = rel * ref.abs() + rel * ref.abs().max()
    err = (got - ref).abs()
    bad = err > tol
    assert not bad.any(), (f"{name}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, max err "
                           f"{err.max().item():.4e}, ref max {ref.abs().max().item():.4e}, first bad idx "
                           f"{bad.nonzero()[0].tolist()}")


@pytest.mark.parametrize("M,N,K_", [(256, 128, 64), (128, 256, 128), (300, 200, 136), (1000, 320, 320),
                                    (154, 64, 768), (2, 1280, 320), (4096, 640, 2560), (70, 36, 72)])
def test_gemm_plain(dev, M, N, K_):
    k = K()
    g = torch.Generator().manual_seed(M * 7 + N)
    a = h(torch.randn(M, K_, generator=g))
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    r = h(torch.randn(M, N, generator=g))
    out = k.gemm(a.half().to(dev), w.half().to(dev))
    close(out, a @ w.T, name="gemm")
    out = k.gemm(a.half().to(dev), w.half().to(dev), b.half().to(dev), residual=r.half().to(dev), out_scale=0.5)
    close(out, (a @ w.T + b + r) * 0.5, name="gemm+bias+residual")


@pytest.mark.parametrize("M,N,K_,kind", [
    (49152, 320, 320, "plain"), (49152, 640, 136, "plain"), (16384, 320, 640, "plain"), (16500, 320, 200, "plain"),
    (49152, 640, 64, "geglu"), (16384, 320, 64, "rowperm"), (49152, 320, 128, "dual"), (320, 40960, 64, "vt"),
    # M tails through the LDS-staged row-contiguous epilogue (K % 64 == 0 keeps them on the 8-wave kernel)
    (16500, 320, 192, "plain"), (33000, 640, 128, "geglu"), (16416, 320, 128, "rowperm"), (40000, 960, 320, "plain")])
def test_gemm_big_tiles(dev, M, N, K_, kind):
    """shapes that take the 8-wave LDS-DMA kernel (N % 320 == 0, >= 128 tiles): 256- and 128-row tiles, M / K tails,
    every epilogue / store mode."""
    k = K()
    g = torch.Generator().manual_seed(M + N + K_)
    a = h(torch.randn(M, K_, generator=g))
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    ad, wd, bd = a.half().to(dev), w.half().to(dev), b.half().to(dev)
    if kind == "plain":
        r = h(torch.randn(M, N, generator=g))
        close(k.gemm(ad, wd, bd, residual=r.half().to(dev)), a @ w.T + b + r, name="big gemm")
    elif kind == "geglu":
        y = a @ w.T + b
        close(k.gemm(ad, wd, bd, epilogue=k.I2V_EPI_GEGLU), y[:, 0::2] * F.gelu(y[:, 1::2]), name="big geglu")
    elif kind == "rowperm":
        B_, F_, HW = 2, 16, M // 32
        r = h(torch.randn(M, N, generator=g))
        y = (a @ w.T + b).reshape(B_, HW, F_, N).permute(0, 2, 1, 3).reshape(M, N) + r
        close(k.gemm(ad, wd, bd, residual=r.half().to(dev), store=k.I2V_STORE_ROWPERM, frames=F_, hw=HW), y,
              name="big rowperm")
    elif kind == "dual":
        a2 = h(torch.randn(M, 64, generator=g))
        w2 = h(torch.randn(N, K_ + 64, generator=g) / 12)
        close(k.gemm(ad, w2.half().to(dev), a2=a2.half().to(dev)), torch.cat([a, a2], 1) @ w2.T, name="big dual")
    else:   # V^T projection: A = weights [C, K], W = tokens [T, K], T = N here
        vt = k.project_vt(wd, ad, 4096)       # tokens = w (N rows), weight = a (M = 320 channels)
        ref = (w @ a.T).reshape(N // 4096, 4096, M).permute(0, 2, 1)
        close(vt[:, :, :4096], ref, name="big vt")


def test_gemm_and_conv_split_k(dev):
    """small M, long K (the 8 x 8 level): K split over workgroups + fp32 reduce kernel with the fused epilogue."""
    k = K()
    g = torch.Generator().manual_seed(23)
    M, N, K_ = 2048, 1280, 5120
    a, w = h(torch.randn(M, K_, generator=g)), h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b, r = h(torch.randn(N, generator=g)), h(torch.randn(M, N, generator=g))
    out = k.gemm(a.half().to(dev), w.half().to(dev), b.half().to(dev), residual=r.half().to(dev))
    close(out, a @ w.T + b + r, name="split-K gemm")
    n, hh, ww, cin, cout = 32, 8, 8, 320, 1280
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    wc = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    bc = h(torch.randn(cout, generator=g))
    rv = h(torch.randn(n, cout, generator=g))
    ref = F.conv2d(x, wc, bc, padding=1) + rv[:, :, None, None]
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(wc).to(dev)
    out = k.conv3x3(xt, wp, bc.half().to(dev), rowvec=rv.half().to(dev), rows_per_vec=hh * ww)
    close(out.permute(0, 3, 1, 2), ref, name="split-K conv")


@pytest.mark.parametrize("M,N,K_,kind", [
    (2048, 1280, 1280, "plain"), (2000, 1280, 640, "plain"), (2048, 2560, 1280, "plain"), (2048, 3840, 1280, "plain"),
    (512, 1280, 1280, "plain"), (128, 1280, 2560, "plain"), (1280, 2048, 1280, "vt"), (640, 2048, 640, "vt"),
    (2048, 1280, 1280, "rowperm"), (2048, 1280, 1280, "dual"), (1100, 384, 704, "plain")])
def test_gemm_deep_pipeline_small_levels(dev, M, N, K_, kind):
    """problems that cannot fill the chip with output tiles and have >= 10 K tiles (the 8 x 8 level: 2048 rows): the 128 x 128
    (four stages) / 128 x 256 (three stages) deep-pipeline form of the 8-wave kernel, every store mode it takes, ragged M,
    N not a multiple of 320, the concatenated second source."""
    k = K()
    g = torch.Generator().manual_seed(M + 3 * N + K_)
    a = h(torch.randn(M, K_, generator=g))
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    ad, wd, bd = a.half().to(dev), w.half().to(dev), b.half().to(dev)
    if kind == "plain":
        r = h(torch.randn(M, N, generator=g))
        close(k.gemm(ad, wd, bd), a @ w.T + b, name="deep gemm")
        close(k.gemm(ad, wd, bd, residual=r.half().to(dev), out_scale=0.5), (a @ w.T + b + r) * 0.5, name="deep gemm + residual")
        rpv = 50 if M % 50 == 0 else 64
        rv = h(torch.randn(M // rpv, N, generator=g))
        close(k.gemm(ad, wd, bd, rowvec=rv.half().to(dev), rows_per_vec=rpv), a @ w.T + b + rv.repeat_interleave(rpv, 0),
              name="deep gemm + row vector")
    elif kind == "rowperm":
        B_, F_, HW = 2, 16, M // 32
        r = h(torch.randn(M, N, generator=g))
        y = (a @ w.T + b).reshape(B_, HW, F_, N).permute(0, 2, 1, 3).reshape(M, N) + r
        close(k.gemm(ad, wd, bd, residual=r.half().to(dev), store=k.I2V_STORE_ROWPERM, frames=F_, hw=HW), y, name="deep rowperm")
    elif kind == "dual":
        a2 = h(torch.randn(M, 640, generator=g))
        w2 = h(torch.randn(N, K_ + 640, generator=g) / math.sqrt(K_ + 640))
        close(k.gemm(ad, w2.half().to(dev), bd, a2=a2.half().to(dev)), torch.cat([a, a2], 1) @ w2.T + b, name="deep dual")
    else:   # V^T projection: A = weights [C, K], W = tokens [T, K], T = N here (64 keys per batch)
        vt = k.project_vt(wd, ad, 64)
        ref = (w @ a.T).reshape(N // 64, 64, M).permute(0, 2, 1)
        close(vt[:, :, :64], ref, name="deep vt")


@pytest.mark.parametrize("batches,L,C,Kd", [(4, 6, 24, 16), (2, 64, 320, 64), (3, 4096, 640, 64), (5, 16, 320, 320)])
def test_gemm_store_vt_t(dev, batches, L, C, Kd):
    """V^T from the natural operand order (A = tokens): generic kernel (small) and 256-row tile kernel (large)."""
    k = K()
    g = torch.Generator().manual_seed(19 + L)
    tok = h(torch.randn(batches * L, Kd, generator=g))
    wv = h(torch.randn(C, Kd, generator=g) / math.sqrt(Kd))
    ld = k.pad8(L)
    out = torch.zeros((batches, C, ld), dtype=torch.float16, device=dev)
    k.gemm(tok.half().to(dev), wv.half().to(dev), store=k.I2V_STORE_VT_T, vt_len=L, vt_ld=ld, out=out)
    ref = (tok @ wv.T).reshape(batches, L, C).permute(0, 2, 1)
    close(out[:, :, :L], ref, name="store VT_T")


@pytest.mark.parametrize("n,hh,ww,cin,cout,stride,up", [(12, 64, 64, 32, 320, 1, False), (12, 32, 32, 64, 320, 1, True),
                                                         (16, 64, 64, 16, 640, 2, False), (4, 64, 64, 8, 320, 1, False),
                                                         # the VAE's channel counts: 256- and 128-column tiles of the 8-wave kernel
                                                         (4, 64, 64, 128, 256, 1, False), (4, 32, 32, 128, 128, 1, True),
                                                         (4, 64, 64, 256, 512, 1, False), (8, 64, 64, 64, 128, 2, False),
                                                         (3, 50, 40, 128, 384, 1, False)])
def test_conv3x3_big_tiles(dev, n, hh, ww, cin, cout, stride, up):
    k = K()
    g = torch.Generator().manual_seed(cin + cout)
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g))
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=1)
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(w).to(dev)
    out = k.conv3x3(xt, wp, b.half().to(dev), stride=stride, upsample=up)
    close(out.permute(0, 3, 1, 2), ref, name="big conv3x3")


@pytest.mark.parametrize("n,hh,ww,cin,cout,size", [(8, 32, 32, 64, 320, (63, 64)), (8, 32, 32, 64, 320, (64, 63)), (4, 16, 16, 640, 640, (31, 31)),
                                                   (3, 7, 5, 16, 40, (13, 10)), (2, 1, 1, 32, 64, (1, 1))])
def test_conv3x3_upsample_to_output_size(dev, n, hh, ww, cin, cout, size):
    """Upsample2D with `output_size` (unet:1414-1415 forward_upsample_size; diffusers Upsample2D: F.interpolate(size=..., mode="nearest")
    then the conv): sizes 2x and 2x - 1, through the 8-wave and the generic kernels."""
    k = K()
    g = torch.Generator().manual_seed(cin + cout + size[0])
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g))
    ref = F.conv2d(F.interpolate(x, size=size, mode="nearest"), w, b, padding=1)
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    out = k.conv3x3(xt, _pack_conv(w).to(dev), b.half().to(dev), upsample=True, output_size=size)
    assert tuple(out.shape) == (n, size[0], size[1], cout)
    close(out.permute(0, 3, 1, 2), ref, name=f"conv3x3 after nearest upsampling to {size}")
    with pytest.raises(NotImplementedError, match="2x"):
        k.conv3x3(xt, _pack_conv(w).to(dev), b.half().to(dev), upsample=True, output_size=(2 * hh + 1, 2 * ww))


def test_gemm_rowvec_gelu(dev):
    k = K()
    g = torch.Generator().manual_seed(3)
    M, N, K_, rpv = 384, 96, 64, 48
    a, w = h(torch.randn(M, K_, generator=g)), h(torch.randn(N, K_, generator=g) / 8)
    b, rv = h(torch.randn(N, generator=g)), h(torch.randn(M // rpv, N, generator=g))
    out = k.gemm(a.half().to(dev), w.half().to(dev), b.half().to(dev), rowvec=rv.half().to(dev), rows_per_vec=rpv)
    close(out, a @ w.T + b + rv.repeat_interleave(rpv, 0), name="gemm+rowvec")
    out = k.gemm(a.half().to(dev), w.half().to(dev), b.half().to(dev), epilogue=k.I2V_EPI_GELU)
    close(out, F.gelu(a @ w.T + b), name="gemm+gelu")


@pytest.mark.parametrize("M,C,I", [(256, 64, 256), (130, 32, 100)])
def test_gemm_geglu(dev, M, C, I):
    k = K()
    g = torch.Generator().manual_seed(5)
    a = h(torch.randn(M, C, generator=g))
    w = h(torch.randn(2 * I, C, generator=g) / math.sqrt(C))
    b = h(torch.randn(2 * I, generator=g))
    y = a @ w.T + b
    ref = y[:, :I] * F.gelu(y[:, I:])
    wi = torch.stack([w[:I], w[I:]], dim=1).reshape(2 * I, C)      # rows interleaved (value_i, gate_i)
    bi = torch.stack([b[:I], b[I:]], dim=1).reshape(2 * I)
    out = k.gemm(a.half().to(dev), wi.half().to(dev), bi.half().to(dev), epilogue=k.I2V_EPI_GEGLU)
    close(out, ref, name="geglu")


def test_gemm_dual_source(dev):
    k = K()
    g = torch.Generator().manual_seed(6)
    M, K1, K2, N = 200, 64, 40, 72
    a1, a2 = h(torch.randn(M, K1, generator=g)), h(torch.randn(M, K2, generator=g))
    w = h(torch.randn(N, K1 + K2, generator=g) / 10)
    out = k.gemm(a1.half().to(dev), w.half().to(dev), a2=a2.half().to(dev))
    close(out, torch.cat([a1, a2], 1) @ w.T, name="dual-source")


def test_gemm_strided_a(dev):
    k = K()
    g = torch.Generator().manual_seed(7)
    big = h(torch.randn(128, 192, generator=g))
    w = h(torch.randn(48, 64, generator=g) / 8)
    bd = big.half().to(dev)
    out = k.gemm(bd[:, 64:128], w.half().to(dev))
    close(out, big[:, 64:128] @ w.T, name="strided A")


@pytest.mark.parametrize("B,F_,HW,C,N", [(2, 4, 16, 64, 64), (1, 16, 64, 32, 40), (2, 8, 9, 64, 128)])
def test_gemm_rowperm(dev, B, F_, HW, C, N):
    """motion-module exit: rows in (b, pixel, frame) order -> (b, frame, pixel) order, residual in the latter."""
    k = K()
    g = torch.Generator().manual_seed(8)
    M = B * F_ * HW
    a, w = h(torch.randn(M, C, generator=g)), h(torch.randn(N, C, generator=g) / 8)
    b, r = h(torch.randn(N, generator=g)), h(torch.randn(M, N, generator=g))
    y = (a @ w.T + b).reshape(B, HW, F_, N).permute(0, 2, 1, 3).reshape(M, N) + r
    out = k.gemm(a.half().to(dev), w.half().to(dev), b.half().to(dev), residual=r.half().to(dev),
                 store=k.I2V_STORE_ROWPERM, frames=F_, hw=HW)
    close(out, y, name="rowperm")


@pytest.mark.parametrize("batches,L,C,Kd", [(3, 64, 64, 32), (2, 77, 40, 768), (5, 16, 320, 320), (4, 6, 24, 16)])
def test_project_vt(dev, batches, L, C, Kd):
    k = K()
    g = torch.Generator().manual_seed(9)
    tok = h(torch.randn(batches * L, Kd, generator=g))
    wv = h(torch.randn(C, Kd, generator=g) / math.sqrt(Kd))
    vt = k.project_vt(tok.half().to(dev), wv.half().to(dev), L)
    ref = (tok @ wv.T).reshape(batches, L, C).permute(0, 2, 1)
    close(vt[:, :, :L], ref, name="project_vt")


@pytest.mark.parametrize("n,hh,ww,cin,cout,stride,up", [
    (2, 8, 8, 8, 32, 1, False), (3, 16, 12, 32, 64, 1, False), (2, 16, 16, 64, 48, 2, False),
    (2, 8, 8, 32, 32, 1, True), (1, 7, 9, 16, 24, 2, False), (2, 32, 32, 320, 320, 1, False),
    (1, 5, 5, 8, 4, 1, False)])
def test_conv3x3(dev, n, hh, ww, cin, cout, stride, up):
    k = K()
    g = torch.Generator().manual_seed(10 + cin)
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g))
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=1)
    oh, ow = ref.shape[-2:]
    rv = h(torch.randn(n, cout, generator=g))
    res = h(torch.randn(n, cout, oh, ow, generator=g))
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(w).to(dev)
    out = k.conv3x3(xt, wp, b.half().to(dev), stride=stride, upsample=up)
    close(out.permute(0, 3, 1, 2), ref, name="conv3x3")
    out = k.conv3x3(xt, wp, b.half().to(dev), stride=stride, upsample=up, rowvec=rv.half().to(dev),
                    rows_per_vec=oh * ow, residual=res.permute(0, 2, 3, 1).contiguous().half().to(dev))
    close(out.permute(0, 3, 1, 2), ref + rv[:, :, None, None] + res, name="conv3x3+temb+res")


@pytest.mark.parametrize("n,hh,ww,cin,cout", [(2, 8, 16, 128, 3), (3, 16, 32, 320, 4), (1, 24, 48, 64, 16), (2, 8, 32, 192, 5),
                                              (2, 64, 64, 320, 4)])
def test_conv3x3_narrow_output(dev, n, hh, ww, cin, cout):
    """3x3 convolutions with <= 16 output channels on images of whole 8 x 16 pixel tiles take the halo-tile kernel
    (csrc/conv_thin.hip: the UNet's conv_out, unet:879-881, 1443; the VAE decoder's): against the exact convolution of the
    fp16-rounded operands, fp32 and fp16 results, image borders and tile seams included (every tile has a border or a seam)."""
    k = K()
    g = torch.Generator().manual_seed(cout * 1000 + cin + hh)
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(w).to(dev)
    out32 = k.conv3x3(xt, wp, b.half().to(dev), out_f32=True)
    out16 = k.conv3x3(xt, wp, b.half().to(dev))
    assert out32.dtype == torch.float32 and out16.dtype == torch.float16 and tuple(out32.shape) == (n, hh, ww, cout)
    close(out32.permute(0, 3, 1, 2), ref, rel=1e-5, name="narrow conv3x3, fp32 result")
    assert torch.equal(out32.half(), out16), "the fp16 form must be the rounding of the fp32 form"
    # a single hot pixel lands in its nine neighbours with the nine taps (orientation of the taps, seams between tiles)
    xi = torch.zeros(1, cin, hh, ww)
    py, px = 7 % hh, 16 % ww              # on a tile seam where the image has more than one tile
    xi[0, :, py, px] = 1.0
    got = k.conv3x3(xi.permute(0, 2, 3, 1).contiguous().half().to(dev), wp, None, out_f32=True).permute(0, 3, 1, 2)
    close(got, F.conv2d(xi.double(), w.double(), None, padding=1).float(), rel=1e-5, name="narrow conv3x3, impulse")


@pytest.mark.parametrize("cout,cin", [(4, 320), (3, 128), (8, 64)])
def test_conv3x3_fp32_result(dev, cout, cin):
    """narrow convolutions can keep their result in fp32 (i2v_gemm_params.c_is_f32): the UNet's 4-channel conv_out
    (unet:879-881) feeding the CFG / DDIM kernel, the VAE decoder's 3-channel image.  Same accumulators as the fp16 form,
    without the final rounding: the error against the exact convolution drops to the fp32 accumulation order."""
    k = K()
    g = torch.Generator().manual_seed(cout * 100 + cin)
    x = h(torch.randn(2, cin, 12, 10, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(w).to(dev)
    out32 = k.conv3x3(xt, wp, b.half().to(dev), out_f32=True)
    out16 = k.conv3x3(xt, wp, b.half().to(dev))
    assert out32.dtype == torch.float32 and out16.dtype == torch.float16
    close(out32.permute(0, 3, 1, 2), ref, rel=1e-5, name="conv3x3 fp32 result")
    assert torch.equal(out32.half(), out16), "the fp16 form must be the rounding of the fp32 form"
    # both edge kernels take the fp32 tokens
    close(k.tokens_to_nchw(out32, dtype=torch.float32), ref, rel=1e-5, name="tokens_to_nchw fp32 source")
    assert torch.equal(k.tokens_to_nchw(out32, dtype=torch.float16), k.tokens_to_nchw(out16))
    with pytest.raises(ValueError):
        k.conv3x3(xt, _pack_conv(h(torch.randn(128, cin, 3, 3, generator=g))).to(dev), out_f32=True)


def _pack_conv(w):
    """conv weights in the contraction order the kernel walks for this channel count (the product's packing function)"""
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
    return pack_conv3x3(w)


def _attn_ref(q, k_, v, heads, group):
    bq, lq, c = q.shape
    d = c // heads
    kk = k_.repeat_interleave(group, 0)
    vv = v.repeat_interleave(group, 0)
    qh = q.view(bq, lq, heads, d).transpose(1, 2)
    kh = kk.view(bq, -1, heads, d).transpose(1, 2)
    vh = vv.view(bq, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(bq, lq, c)


@pytest.mark.parametrize("bq,group,heads,d,lq,lk", [
    (2, 1, 2, 8, 64, 64), (4, 2, 4, 16, 100, 100), (2, 1, 2, 32, 256, 256), (8, 4, 8, 40, 256, 256),
    (2, 1, 8, 40, 1024, 1024), (2, 1, 2, 64, 130, 77), (2, 1, 8, 80, 256, 77), (2, 2, 4, 160, 64, 64),
    (1, 1, 2, 160, 300, 200), (3, 1, 2, 24, 16, 4), (2, 1, 3, 48, 40, 129), (2, 1, 2, 96, 128, 64),
    (2, 1, 1, 128, 128, 192), (2, 1, 2, 40, 200, 700), (2, 2, 2, 64, 130, 513), (1, 1, 2, 16, 64, 640)])
def test_attention(dev, bq, group, heads, d, lq, lk):
    k = K()
    g = torch.Generator().manual_seed(bq * 100 + d + lq)
    c = heads * d
    bkv = bq // group
    q = h(torch.randn(bq, lq, c, generator=g))
    kk = h(torch.randn(bkv, lk, c, generator=g))
    v = h(torch.randn(bkv, lk, c, generator=g))
    ref = _attn_ref(q, kk, v, heads, group)
    ld = k.pad8(lk)
    vt = torch.full((bkv, c, ld), float("nan"))           # tail keys hold garbage: the kernel must mask them
    vt[:, :, :lk] = v.permute(0, 2, 1)
    out = k.attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                      batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    close(out.view(bq, lq, c), ref, name="attention")
    prev = h(torch.randn(bq * lq, c, generator=g))
    out2 = prev.half().to(dev)
    k.attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group, out=out2, accumulate=True,
                acc_scale=0.75)
    close(out2.view(bq, lq, c), prev.view(bq, lq, c) + 0.75 * ref, name="attention accumulate")


@pytest.mark.parametrize("bq,group,heads,d,lq,lk,amp", [
    (2, 1, 8, 40, 256, 512, 2.0), (4, 2, 8, 40, 256, 512, 4.0), (2, 1, 8, 40, 128, 64, 4.0), (2, 1, 4, 80, 128, 77, 3.0),
    (2, 1, 2, 160, 64, 192, 4.0), (2, 1, 2, 64, 130, 320, 3.0), (2, 2, 2, 48, 100, 200, 4.0), (1, 1, 2, 128, 64, 128, 3.0)])
def test_attention_large_logits(dev, bq, group, heads, d, lq, lk, amp):
    """q and k scaled so that logits reach 20 - 90 (trained attention layers do; unit-variance inputs stop near 6).  The
    running max must be the maximum over ALL keys of the tile: taken from one lane group only (rounds 1-2, a folded
    cross-lane reduction) the result stays exact while exp2(s - m) fits fp16 and saturates / overflows beyond -- weights of
    the large keys clipped, NaN from the kernels that sum P on the VALU.  Tolerance: the fp16 rounding of the pre-scaled Q
    moves a logit by ~|logit| 2^-11."""
    k = K()
    g = torch.Generator().manual_seed(int(amp * 10) + d + lk)
    c = heads * d
    bkv = bq // group
    q = h(torch.randn(bq, lq, c, generator=g) * amp)
    kk = h(torch.randn(bkv, lk, c, generator=g) * amp)
    v = h(torch.randn(bkv, lk, c, generator=g))
    ref = _attn_ref(q, kk, v, heads, group)
    ld = k.pad8(lk)
    vt = torch.zeros((bkv, c, ld))
    vt[:, :, :lk] = v.permute(0, 2, 1)
    out = k.attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                      batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    close(out.view(bq, lq, c), ref, rel=6e-4 * amp * amp, name=f"attention, logits x{amp * amp:g}")


def test_attention_spikes_in_every_lane_group(dev):
    """one key far above the rest per (query tile, lane group): keys 8 g + 3 (+ 32, + a later tile) belong to lane group g of the
    S^T accumulator; each group's spike must move the running max of the whole row."""
    k = K()
    g = torch.Generator().manual_seed(5)
    bq, heads, d, lq, lk = 4, 2, 40, 128, 384
    c = heads * d
    q = h(torch.randn(bq, lq, c, generator=g))
    kk = h(torch.randn(bq, lk, c, generator=g))
    v = h(torch.randn(bq, lk, c, generator=g))
    for b in range(bq):                       # batch b: the spike sits in lane group b, in key tile b + 1
        kk[b, 64 * (b + 1) + 8 * b + 3 + 32 * (b & 1)] *= 14.0
    ref = _attn_ref(q, kk, v, heads, 1)
    vt = v.permute(0, 2, 1).contiguous()
    out = k.attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                      batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d)
    close(out.view(bq, lq, c), ref, rel=4e-3, name="attention, spiked keys")


def test_attention_config5_key_length(dev):
    """BASELINE config 5's L0 level: Lq = Lk = 9216 (96 x 96 latent), head_dim 40, in the self (kv_group 1) and the
    cross-frame (every frame reads frame 0's K / V) forms, against torch SDPA on the host.  144 key tiles per query
    block: the online-softmax rescale chain at a length the UNet-level property tests never compared with anything."""
    import torch.nn.functional as F
    k = K()
    g = torch.Generator().manual_seed(9216)
    heads, d, L = 8, 40, 9216
    c = heads * d
    for bq, group in ((1, 1), (2, 2)):
        bkv = bq // group
        q = h(torch.randn(bq, L, c, generator=g))
        kk = h(torch.randn(bkv, L, c, generator=g))
        v = h(torch.randn(bkv, L, c, generator=g))
        sp = lambda t: t.view(t.shape[0], L, heads, d).transpose(1, 2)
        ref = F.scaled_dot_product_attention(sp(q), sp(kk.repeat_interleave(group, 0)), sp(v.repeat_interleave(group, 0)))
        ref = ref.transpose(1, 2).reshape(bq, L, c)
        vt = v.permute(0, 2, 1).contiguous()
        out = k.attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                          batch_q=bq, lq=L, lk=L, heads=heads, head_dim=d, kv_group=group)
        close(out.view(bq, L, c), ref, name=f"attention Lq=Lk=9216 d=40 group={group}")


def _variants_env(**switches):
    """environment of a child process that runs the measured-and-rejected kernel forms (csrc/variants/, the 4-wave GEMM
    instantiations): they are not in the default library, only in the A/B build `bash tools/build_variant.sh --variants`
    (-> .ab_libs/variants.so, selected through I2V_LIB_PATH).  Skips when that library has not been built."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, ".ab_libs", "variants.so")
    if not os.path.exists(lib):
        pytest.skip("variant kernels are not in the default build: bash tools/build_variant.sh --variants")
    # a variants library built before the last ABI change (it is not rebuilt by __graft_entry__.build()) is stale, not wrong
    import ctypes
    import i2v_adapter_unofficial_amd as pkg
    handle = ctypes.CDLL(lib)
    missing = [n for n in pkg._lib.SIGNATURES if not hasattr(handle, n)]
    if missing or handle.i2v_abi_version() != pkg._lib.ABI_VERSION:
        pytest.skip(f".ab_libs/variants.so is older than include/i2v_hip.h (missing {missing[:3]}): rebuild it with "
                    "bash tools/build_variant.sh --variants")
    # ... and one that was not built from THIS tree's kernel sources proves nothing about them (tools/build_variant.sh writes the
    # hash of csrc/ + the header beside the library)
    stamp_file = lib + ".stamp"
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    headers = [os.path.join(ge.CSRC, f) for f in os.listdir(ge.CSRC) if f.endswith(".h")] + [os.path.join(root, "include", "i2v_hip.h")]
    if not os.path.exists(stamp_file) or open(stamp_file).read().strip() != ge._stamp(headers):
        pytest.skip(".ab_libs/variants.so was not built from this tree's kernel sources: bash tools/build_variant.sh --variants")
    return dict(os.environ, I2V_LIB_PATH=lib, **switches)


@pytest.mark.variants
def test_attention_32x32_formulation_opt_in(dev):
    """the head_dim-40 kernel on 32x32x16 MFMAs (csrc/variants/attention32.hip) is opt-in (I2V_ATTN32=1, read once per process): run
    it in a child process against torch SDPA, with a query tail, a key tail, K/V sharing and the accumulate form."""
    import os, subprocess, sys
    code = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for bq, group, heads, lq, lk in ((4, 2, 8, 300, 700), (2, 1, 8, 1024, 1024)):
    d = 40; c = heads * d; bkv = bq // group
    hf = lambda t: t.half().float()
    q, kk, v = hf(torch.randn(bq, lq, c, generator=g)), hf(torch.randn(bkv, lk, c, generator=g)), hf(torch.randn(bkv, lk, c, generator=g))
    sp = lambda t, b: t.view(b, -1, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, bq), sp(kk.repeat_interleave(group, 0), bq), sp(v.repeat_interleave(group, 0), bq)).transpose(1, 2).reshape(bq, lq, c)
    ld = k.pad8(lk); vt = torch.full((bkv, c, ld), float("nan")); vt[:, :, :lk] = v.permute(0, 2, 1)
    args = (q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev))
    kw = dict(batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    out = k.attention(*args, **kw).float().cpu().view(bq, lq, c)
    err = (out - ref).abs().max().item()
    prev = hf(torch.randn(bq * lq, c, generator=g)); out2 = prev.half().to(dev)
    k.attention(*args, out=out2, accumulate=True, acc_scale=0.75, **kw)
    err2 = (out2.float().cpu().view(bq, lq, c) - (prev.view(bq, lq, c) + 0.75 * ref)).abs().max().item()
    print("ERR", err, err2)
    assert err < 3e-3 and err2 < 5e-3, (err, err2)
print("OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=_variants_env(I2V_ATTN32="1"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.variants
def test_attention_software_pipelined_opt_in_child_process(dev):
    """I2V_ATTN_PIPE=1 (read once per process, hence the child): the three-stage software-pipelined key loop of csrc/variants/attention_pipe.hip
    for head_dim 40 and whole 64-key tiles -- same results as the default kernel at unit and at large logits, odd and even tile
    counts, cross-frame groups, accumulate; measured slower (profiles/r3_attn_pipe_ab.txt), kept as a tested A/B switch."""
    import os, subprocess, sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import torch.nn.functional as F
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(4)
for bq, group, heads, lq, lk, amp in ((4, 2, 8, 300, 704, 1.0), (2, 1, 8, 1024, 1024, 1.0), (2, 1, 4, 256, 192, 3.0), (2, 2, 2, 128, 320, 4.0)):
    d = 40; c = heads * d; bkv = bq // group
    hf = lambda t: t.half().float()
    q, kk, v = hf(torch.randn(bq, lq, c, generator=g) * amp), hf(torch.randn(bkv, lk, c, generator=g) * amp), hf(torch.randn(bkv, lk, c, generator=g))
    sp = lambda t, b: t.view(b, -1, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, bq), sp(kk.repeat_interleave(group, 0), bq), sp(v.repeat_interleave(group, 0), bq)).transpose(1, 2).reshape(bq, lq, c)
    vt = v.permute(0, 2, 1).contiguous()
    args = (q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev))
    kw = dict(batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    out = k.attention(*args, **kw).float().cpu().view(bq, lq, c)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    prev = hf(torch.randn(bq * lq, c, generator=g)); out2 = prev.half().to(dev)
    k.attention(*args, out=out2, accumulate=True, acc_scale=0.75, **kw)
    err2 = (out2.float().cpu().view(bq, lq, c) - (prev.view(bq, lq, c) + 0.75 * ref)).abs().max().item()
    print("ERR", err, err2)
    assert err < 1e-3 * amp * amp and err2 < 5e-3 * amp * amp, (err, err2)
print("OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=_variants_env(I2V_ATTN_PIPE="1"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_attention_strided_qk_and_spike(dev):
    """q / k read as column slices of a fused projection; one spiked key forces the online-softmax rescale."""
    k = K()
    g = torch.Generator().manual_seed(77)
    bq, heads, d, lq, lk = 2, 4, 40, 192, 320
    c = heads * d
    qk = h(torch.randn(bq * lq, 2 * c, generator=g))
    kv_src = h(torch.randn(bq * lk, 2 * c, generator=g))
    kv_src[lk // 2 + 70, c:] *= 12.0      # a key far above the running max, in the 3rd key tile
    v = h(torch.randn(bq, lk, c, generator=g))
    q, kk = qk[:, :c].reshape(bq, lq, c), kv_src[:, c:].reshape(bq, lk, c)
    ref = _attn_ref(q, kk, v, heads, 1)
    vt = v.permute(0, 2, 1).contiguous()
    qd, kd = qk.half().to(dev), kv_src.half().to(dev)
    out = k.attention(qd[:, :c], kd[:, c:], vt.half().to(dev), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d)
    close(out.view(bq, lq, c), ref, name="attention strided+spike")


@pytest.mark.parametrize("npix,frames,heads,d", [(10, 16, 8, 40), (7, 8, 4, 8), (5, 4, 2, 16), (33, 32, 8, 80),
                                                 (3, 24, 2, 160), (64, 16, 8, 160), (9, 5, 3, 24)])
def test_temporal_attention(dev, npix, frames, heads, d):
    k = K()
    g = torch.Generator().manual_seed(npix + frames)
    c = heads * d
    q = h(torch.randn(npix, frames, c, generator=g))
    kk = h(torch.randn(npix, frames, c, generator=g))
    v = h(torch.randn(npix, frames, c, generator=g))
    ref = _attn_ref(q, kk, v, heads, 1)
    ld = k.pad8(frames)
    vt = torch.full((npix, c, ld), float("nan"))
    vt[:, :, :frames] = v.permute(0, 2, 1)
    out = k.temporal_attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                               n_pixels=npix, frames=frames, heads=heads, head_dim=d)
    close(out.view(npix, frames, c), ref, name="temporal attention")


@pytest.mark.parametrize("npix,frames,heads,d,amp", [(64, 16, 8, 40, 3.0), (33, 32, 8, 80, 4.0), (8192, 16, 8, 40, 4.0),
                                                     (16, 24, 2, 160, 3.0)])
def test_temporal_attention_large_logits(dev, npix, frames, heads, d, amp):
    """the frame-axis attention with logits up to ~90 (both kernels: one wave per (pixel, head) and the LDS-staged 64 x 64 form)"""
    k = K()
    g = torch.Generator().manual_seed(npix + frames + 1)
    c = heads * d
    q = h(torch.randn(npix, frames, c, generator=g) * amp)
    kk = h(torch.randn(npix, frames, c, generator=g) * amp)
    v = h(torch.randn(npix, frames, c, generator=g))
    ref = _attn_ref(q, kk, v, heads, 1)
    vt = torch.zeros((npix, c, k.pad8(frames)))
    vt[:, :, :frames] = v.permute(0, 2, 1)
    out = k.temporal_attention(q.reshape(-1, c).half().to(dev), kk.reshape(-1, c).half().to(dev), vt.half().to(dev),
                               n_pixels=npix, frames=frames, heads=heads, head_dim=d)
    close(out.view(npix, frames, c), ref, rel=6e-4 * amp * amp, name="temporal attention, large logits")


@pytest.mark.parametrize("npix,amp,strided,frames", [(8, 1.0, False, 16), (1000, 1.0, True, 16), (256, 3.0, False, 16),
                                                      (8 * 771, 1.0, False, 16), (16, 1.0, False, 8), (16 * 37, 3.0, True, 8),
                                                      (4, 1.0, False, 32), (4 * 53, 3.0, True, 32)])
def test_motion_attention_sub_block_fused(dev, npix, amp, strided, frames):
    """i2v_motion_attn_f16: LayerNorm + positional table, q / k / v projections and the attention over the frames of a
    pixel in one launch (channels 320, 8 heads of 40: the SD-1.5 64^2 level) against fp32 torch on the same fp16-rounded
    operands, and against the un-fused kernels it replaces (LayerNorm -> q|k GEMM, V GEMM -> temporal attention).  16 frames (a
    16-row MFMA tile is one pixel's sequence), 8 (two pixels per tile: the scores between them are masked) and 32 (a pixel is
    two tiles: the configurations of 8 f x 256^2 and 32 f x 768^2)."""
    k = K()
    c, heads, d, eps = 320, 8, 40, 1e-5
    rows = npix * frames
    assert k.motion_attn_supported(rows, c, heads, d, frames) and not k.motion_attn_supported(rows + 16, c, heads, d, frames)
    assert not k.motion_attn_supported(rows, 640, 8, 80, frames) and not k.motion_attn_supported(rows, c, heads, d, 4)
    assert not k.motion_attn_supported(rows, c, heads, d, 24)
    g = torch.Generator().manual_seed(npix)
    ld = c + 64 if strided else c
    xb = h(torch.randn(rows, ld, generator=g) * 1.5 + 0.3)
    x = xb[:, :c]
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    pe = h(torch.randn(32, c, generator=g))
    wq, wk, wv = (h(torch.randn(c, c, generator=g) * amp * c ** -0.5) for _ in range(3))
    wv = wv / amp
    n = h(F.layer_norm(x, (c,), gamma, beta, eps) + pe[:frames].repeat(npix, 1))         # (the kernel rounds n to fp16)
    q, kk, v = h(n @ wq.T), h(n @ wk.T), h(n @ wv.T)
    ref = _attn_ref(q.view(npix, frames, c), kk.view(npix, frames, c), v.view(npix, frames, c), heads, 1).reshape(rows, c)
    D = lambda t: t.half().to(dev)
    w = k.pack_motion_qkv(D(wq), D(wk), D(wv), heads)
    assert tuple(w.shape) == (8 * 3 * 48, c)
    xd = D(xb)[:, :c]
    g32, s32 = k.motion_attn_tables(D(gamma), D(beta), D(pe), frames)
    assert g32.dtype == s32.dtype == torch.float32 and tuple(s32.shape) == (frames, c)
    out = k.motion_attn(xd, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps)
    close(out, ref, rel=3e-3 * amp * amp, name="fused motion attention vs fp32 torch")
    # the kernels it replaces, on the same device operands
    nl = k.layernorm(xd.contiguous(), D(gamma), D(beta), eps, pe=D(pe), pe_period=frames)
    qk = k.gemm(nl, D(torch.cat([wq, wk], 0)))
    vv = k.gemm(nl, D(wv))
    old = k.temporal_attention(qk[:, :c], qk[:, c:], k.transpose_tokens(vv, frames), n_pixels=npix, frames=frames, heads=heads,
                               head_dim=d)
    close(out, old, rel=1.5e-3 * amp * amp, name="fused motion attention vs the un-fused kernels")
    assert torch.equal(out, k.motion_attn(xd, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps))
    with pytest.raises(Exception, match="not a fused shape"):
        k.motion_attn(xd[:rows - 16], g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps)
    # to_out + bias + residual in the same launch (w_o, b_o): against fp32 torch on the kernel's own fp16 o, and against the GEMM
    # launch it replaces (fp32 sums in another order: an fp16 ulp of the result)
    wo, bo = h(torch.randn(c, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(c, generator=g))
    op = k.pack_attn_out(D(wo), D(bo), heads)
    assert tuple(op[0].shape) == (8 * 48, c) and op[1].dtype == torch.float32
    full = k.motion_attn(xd, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps, out_proj=op)
    ref_o = x + out.float().cpu() @ wo.T + bo
    close(full, ref_o, rel=1e-3, name="fused motion attention + to_out + residual vs fp32 torch")
    pair = k.gemm(out, D(wo), D(bo), residual=xd)
    close(full, pair, rel=1e-3, name="to_out inside the launch vs the GEMM behind it")
    assert torch.equal(full, k.motion_attn(xd, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps, out_proj=op))
    buf = xd.contiguous().clone()                       # in place: a tile's rows are read before they are written
    assert torch.equal(full, k.motion_attn(buf, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps, out_proj=op, out=buf))
    with pytest.raises(ValueError, match="pack_attn_out"):
        k.motion_attn(xd, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=eps, out_proj=(op[0][:-16], op[1]))


@pytest.mark.parametrize("n_img,L,adapter,strided,amp", [(1, 128, True, False, 1.0), (3, 1024, True, True, 1.0), (2, 4096, False, False, 1.0),
                                                          (5, 256, True, False, 30.0)])
def test_layernorm_qkv_projection_fused(dev, n_img, L, adapter, strided, amp):
    """i2v_ln_qkv_f16: LayerNorm 1, [q | k | q_adapter] (row-major) and V^T (per image [channel][key]) of the spatial block's
    self- / cross-frame attention in one launch (C = 320: the SD-1.5 64^2 level), against fp32 torch on the same fp16-rounded
    operands and against the kernels it replaces (LayerNorm -> i2v_gemm_f16, project_vt).  amp: rows with a large common offset
    (mean / std = 30): the statistics are exact two-pass ones."""
    k = K()
    c, eps = 320, 1e-5
    rows, n_qk = n_img * L, (3 if adapter else 2) * c
    assert k.ln_qkv_supported(rows, c, n_qk, L) and not k.ln_qkv_supported(rows, 640, 3 * 640, L) and not k.ln_qkv_supported(rows, c, 4 * c, L)
    assert not k.ln_qkv_supported(rows + 16, c, n_qk, L) and not k.ln_qkv_supported(rows, c, n_qk, L + 64)
    g = torch.Generator().manual_seed(rows + n_qk)
    ld = c + 64 if strided else c
    xb = h(torch.randn(rows, ld, generator=g) * 1.3 + amp * 0.4)
    x = xb[:, :c]
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    w_qk, w_v = h(torch.randn(n_qk, c, generator=g) * c ** -0.5), h(torch.randn(c, c, generator=g) * c ** -0.5)
    n = h(F.layer_norm(x, (c,), gamma, beta, eps))
    ref_qk = n @ w_qk.T
    ref_vt = (n @ w_v.T).view(n_img, L, c).transpose(1, 2)
    D = lambda t: t.half().to(dev)
    wp = k.pack_ln_qkv(D(w_qk), D(w_v))
    xd = D(xb)[:, :c]
    qk, vt = k.ln_qkv(xd, D(gamma).float(), D(beta).float(), wp, n_qk=n_qk, rows_per_image=L, eps=eps)
    assert tuple(qk.shape) == (rows, n_qk) and tuple(vt.shape) == (n_img, c, L)
    close(qk, ref_qk, rel=3e-3, name="fused LayerNorm + q | k | q_adapter vs fp32 torch")
    close(vt, ref_vt, rel=3e-3, name="fused LayerNorm + V^T vs fp32 torch")
    nl = k.layernorm(xd.contiguous(), D(gamma), D(beta), eps)
    close(qk, k.gemm(nl, D(w_qk)), rel=1e-3, name="fused q | k | q_adapter vs LayerNorm -> GEMM")
    close(vt, k.project_vt(nl, D(w_v), L), rel=1e-3, name="fused V^T vs LayerNorm -> project_vt")
    qk2, vt2 = k.ln_qkv(xd, D(gamma).float(), D(beta).float(), wp, n_qk=n_qk, rows_per_image=L, eps=eps)
    assert torch.equal(qk, qk2) and torch.equal(vt, vt2)
    with pytest.raises(Exception, match="not a fused shape"):          # images of 64 rows: not whole 128-row tiles
        k.ln_qkv(xd, D(gamma).float(), D(beta).float(), wp, n_qk=n_qk, rows_per_image=64, eps=eps)


@pytest.mark.parametrize("clips,frames,L", [(2, 4, 128), (3, 16, 256), (1, 2, 4096)])
def test_layernorm_k0_v0t_of_the_frame0_rows_fused(dev, clips, frames, L):
    """i2v_ln_qkv_f16 with n_qk = C and x_image_stride: LayerNorm 1 of the FRAME-0 rows of every clip (read in place from the
    [clips x frames x L, C] token matrix), the adapter's K0 (row-major) and V0^T in one launch (i2v:484-492) -- against fp32 torch
    and against the three launches it replaces (batched LayerNorm over the row blocks -> GEMM, project_vt)."""
    k = K()
    c, eps = 320, 1e-5
    g = torch.Generator().manual_seed(clips * frames + L)
    x = h(torch.randn(clips * frames * L, c, generator=g) * 1.3 + 0.4)
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    w_k, w_v = h(torch.randn(c, c, generator=g) * c ** -0.5), h(torch.randn(c, c, generator=g) * c ** -0.5)
    first = x.view(clips, frames * L, c)[:, :L].reshape(-1, c)
    n = h(F.layer_norm(first, (c,), gamma, beta, eps))
    D = lambda t: t.half().to(dev)
    xd = D(x)
    k0, v0t = k.ln_qkv(xd, D(gamma).float(), D(beta).float(), k.pack_ln_qkv(D(w_k), D(w_v)), n_qk=c, rows_per_image=L, eps=eps,
                       images=clips, x_image_stride=frames * L * c)
    assert tuple(k0.shape) == (clips * L, c) and tuple(v0t.shape) == (clips, c, L)
    close(k0, n @ w_k.T, rel=3e-3, name="fused frame-0 LayerNorm + K0 vs fp32 torch")
    close(v0t, (n @ w_v.T).view(clips, L, c).transpose(1, 2), rel=3e-3, name="fused frame-0 LayerNorm + V0^T vs fp32 torch")
    nl = k.layernorm(xd.view(clips, frames * L, c)[:, :L], D(gamma), D(beta), eps).reshape(-1, c)
    close(k0, k.gemm(nl, D(w_k)), rel=1e-3, name="fused K0 vs LayerNorm -> GEMM")
    close(v0t, k.project_vt(nl, D(w_v), L), rel=1e-3, name="fused V0^T vs LayerNorm -> project_vt")


@pytest.mark.parametrize("rows,strided", [(128, False), (128 * 300, True), (128 * 771, False)])
def test_feed_forward_fused(dev, rows, strided):
    """i2v_ff_fused_f16: x + W2 GEGLU(LayerNorm(x) W1^T + b1) + b2 in one launch (C = 320, inner 1280: the SD-1.5 64^2 level)
    against fp32 torch on the same fp16-rounded operands and against the un-fused pair (LayerNorm -> GEGLU GEMM -> GEMM + residual)."""
    k = K()
    c, inner, eps = 320, 1280, 1e-5
    assert k.ff_fused_supported(rows, c, inner) and not k.ff_fused_supported(rows + 16, c, inner) and not k.ff_fused_supported(rows, 640, 2560)
    g = torch.Generator().manual_seed(rows)
    ld = c + 32 if strided else c
    xb = h(torch.randn(rows, ld, generator=g) * 1.2 + 0.2)
    x = xb[:, :c]
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    w1, b1 = h(torch.randn(2 * inner, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(2 * inner, generator=g))
    w2, b2 = h(torch.randn(c, inner, generator=g) * inner ** -0.5), h(0.1 * torch.randn(c, generator=g))
    n = h(F.layer_norm(x, (c,), gamma, beta, eps))
    pre = n @ w1.T + b1
    hh = h(pre[:, :inner] * F.gelu(pre[:, inner:]))
    ref = x + hh @ w2.T + b2
    D = lambda t: t.half().to(dev)
    packed = k.pack_ff_fused(D(w1), D(b1), D(w2), D(b2))
    xd = D(xb)[:, :c]
    out = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps)
    close(out, ref, rel=3e-3, name="fused feed-forward vs fp32 torch")
    from i2v_adapter_unofficial_amd.blocks import pack_geglu
    nl = k.layernorm(xd.contiguous(), D(gamma), D(beta), eps)
    w1p, b1p = pack_geglu(D(w1), D(b1))
    old = k.gemm(k.gemm(nl, w1p, b1p, epilogue=k.I2V_EPI_GEGLU), D(w2), D(b2), residual=xd.contiguous())
    close(out, old, rel=1.5e-3, name="fused feed-forward vs the un-fused pair")
    assert torch.equal(out, k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps))
    inplace = xd.contiguous().clone()            # out may alias x: a tile's rows are read (twice) and written by one workgroup only
    k.ff_fused(inplace, D(gamma).float(), D(beta).float(), packed, eps=eps, out=inplace)
    assert torch.equal(inplace, out)
    with pytest.raises(Exception, match="not a fused shape"):
        k.ff_fused(xd[:rows - 16], D(gamma).float(), D(beta).float(), packed, eps=eps)


@pytest.mark.parametrize("batch,frames,hw,strided", [(1, 0, 0, False), (3, 0, 0, True), (1, 16, 8, False), (2, 16, 200, True),
                                                     (1, 8, 48, False), (2, 32, 36, False)])
def test_feed_forward_fused_with_proj_out_tail(dev, batch, frames, hw, strided):
    """the tail of i2v_ff_fused_f16 (ABI 8): the Linear that follows the block in the same launch --
    out[perm(r)] = res2[perm(r)] + fp16(x + FF(LayerNorm(x)))[r] W3^T + b3 -- without a permutation (the spatial transformer's
    proj_out, i2v:298-314) and with rows arriving in (batch, pixel, frame) order and leaving in (batch, frame, pixel) order (the
    motion module's): against fp32 torch on the same fp16-rounded operands, and bit-for-bit against the two launches it replaces
    (fused feed-forward, then i2v_gemm_f16 with the residual / I2V_STORE_ROWPERM)... up to the GEMM kernels' own summation order."""
    k = K()
    c, inner, eps = 320, 1280, 1e-5
    rows = batch * frames * hw if frames else 128 * 5 * batch
    assert rows % 128 == 0
    assert k.ff_fused_tail_supported(rows, c, inner, frames, hw)
    assert not k.ff_fused_tail_supported(rows, c, inner, 12, 32) and not k.ff_fused_tail_supported(rows + 16, c, inner, 0, 0)
    g = torch.Generator().manual_seed(rows + frames)
    ld = c + 32 if strided else c
    xb = h(torch.randn(rows, ld, generator=g) * 1.2 + 0.2)
    x = xb[:, :c]
    res2 = h(torch.randn(rows, c, generator=g))                      # rows in OUTPUT order
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    w1, b1 = h(torch.randn(2 * inner, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(2 * inner, generator=g))
    w2, b2 = h(torch.randn(c, inner, generator=g) * inner ** -0.5), h(0.1 * torch.randn(c, generator=g))
    w3, b3 = h(torch.randn(c, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(c, generator=g))
    n = h(F.layer_norm(x, (c,), gamma, beta, eps))
    pre = n @ w1.T + b1
    y = h(x + h(pre[:, :inner] * F.gelu(pre[:, inner:])) @ w2.T + b2)
    z = y @ w3.T + b3
    if frames:                                                       # (b, pixel, frame) -> (b, frame, pixel)
        z = z.view(batch, hw, frames, c).permute(0, 2, 1, 3).reshape(rows, c)
    ref = res2 + z
    D = lambda t: t.half().to(dev)
    packed = k.pack_ff_fused(D(w1), D(b1), D(w2), D(b2))
    tail = (k.pack_ff_tail(D(w3), D(b3)), D(res2), frames, hw)
    xd = D(xb)[:, :c]
    out = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=tail)
    close(out, ref, rel=3e-3, name="fused feed-forward + proj_out vs fp32 torch")
    yd = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps)
    close(yd, y, rel=3e-3, name="fused feed-forward (no tail) vs fp32 torch")
    store = dict(store=k.I2V_STORE_ROWPERM, frames=frames, hw=hw) if frames else {}
    old = k.gemm(yd, D(w3), D(b3), residual=D(res2), **store)
    close(out, old, rel=1e-3, name="fused feed-forward + proj_out vs the two launches")
    assert torch.equal(out, k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=tail))
    if not frames:         # without a permutation out may alias x and res2
        buf = xd.contiguous().clone()
        k.ff_fused(buf, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=tail, out=buf)
        assert torch.equal(buf, out)
    else:
        with pytest.raises(Exception, match="alias"):
            buf = xd.contiguous().clone()
            k.ff_fused(buf, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=tail, out=buf)


@pytest.mark.parametrize("rows,n_ctx,lt,amp", [(256, 2, 77, 1.0), (128 * 37, 1, 77, 1.0), (1024, 2, 80, 3.0), (512, 4, 5, 1.0)])
def test_text_cross_attention_sub_block_fused(dev, rows, n_ctx, lt, amp):
    """i2v_cross_attn_fused_f16: LayerNorm, to_q and the attention against a <= 80-token context whose K / V^T are given, in one
    launch (channels 320, 8 heads of 40) against fp32 torch on the same fp16-rounded operands, and against the un-fused kernels
    (LayerNorm -> q GEMM -> i2v_attention_f16 with kv_group).  The pad columns of V^T hold NaN: they must not be read as data."""
    k = K()
    c, heads, d, eps = 320, 8, 40, 1e-5
    rpc = rows // n_ctx
    assert k.cross_attn_fused_supported(rows, c, heads, d, lt, rpc) and not k.cross_attn_fused_supported(rows, c, heads, d, 81, rpc)
    assert not k.cross_attn_fused_supported(rows, c, heads, d, lt, rpc + 16) and not k.cross_attn_fused_supported(rows, 640, 8, 80, lt, rpc)
    g = torch.Generator().manual_seed(rows + lt)
    x = h(torch.randn(rows, c, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    wq = h(torch.randn(c, c, generator=g) * amp * c ** -0.5)
    kk = h(torch.randn(n_ctx, lt, c, generator=g) * amp)
    vv = h(torch.randn(n_ctx, lt, c, generator=g))
    n = h(F.layer_norm(x, (c,), gamma, beta, eps))
    q = h(n @ wq.T)
    ref = _attn_ref(q.view(n_ctx, rpc, c), kk, vv, heads, 1).reshape(rows, c)
    D = lambda t: t.half().to(dev)
    ld = k.pad8(lt)
    vt = torch.full((n_ctx, c, ld), float("nan"))
    vt[:, :, :lt] = vv.permute(0, 2, 1)
    w = k.pack_cross_q(D(wq), heads)
    assert tuple(w.shape) == (8 * 48, c)
    g32, b32 = D(gamma).float(), D(beta).float()
    kd, vtd, xd = D(kk.reshape(-1, c)), D(vt), D(x)
    frag = k.pack_ctx_fragments(kd, vtd, heads, lt)
    assert tuple(frag.shape) == (n_ctx, heads, 30, 64, 4) and torch.isfinite(frag.float()).all()
    out = k.cross_attn_fused(xd, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc, eps=eps)
    close(out, ref, rel=3e-3 * amp * amp, name="fused text cross-attention vs fp32 torch")
    nl = k.layernorm(xd, D(gamma), D(beta), eps)
    old = k.attention(k.gemm(nl, D(wq)), kd, vtd, batch_q=n_ctx, lq=rpc, lk=lt, heads=heads, head_dim=d)
    close(out, old, rel=1.5e-3 * amp * amp, name="fused text cross-attention vs the un-fused kernels")
    assert torch.equal(out, k.cross_attn_fused(xd, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc, eps=eps))
    # + the IP-Adapter's decoupled image cross-attention (4 image tokens, weight 0.7): a second softmax over its own K / V
    li, ip_scale = 4, 0.7
    ki, vi = h(torch.randn(n_ctx, li, c, generator=g) * amp), h(torch.randn(n_ctx, li, c, generator=g))
    ref_ip = ref + ip_scale * _attn_ref(q.view(n_ctx, rpc, c), ki, vi, heads, 1).reshape(rows, c)
    vti = torch.full((n_ctx, c, k.pad8(li)), float("nan"))
    vti[:, :, :li] = vi.permute(0, 2, 1)
    frag_ip = k.pack_ctx_fragments(D(ki.reshape(-1, c)), D(vti), heads, li)
    out_ip = k.cross_attn_fused(xd, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc, eps=eps,
                                ip_frag=frag_ip, ip_len=li, ip_scale=ip_scale)
    close(out_ip, ref_ip, rel=3e-3 * amp * amp, name="fused text + image cross-attention vs fp32 torch")
    old_ip = old.clone()
    k.attention(k.gemm(nl, D(wq)), D(ki.reshape(-1, c)), D(vti), batch_q=n_ctx, lq=rpc, lk=li, heads=heads, head_dim=d, out=old_ip,
                accumulate=True, acc_scale=ip_scale)
    close(out_ip, old_ip, rel=1.5e-3 * amp * amp, name="fused text + image cross-attention vs the un-fused kernels")
    with pytest.raises(Exception, match="not a fused shape"):
        k.cross_attn_fused(xd[:rows - 16], g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc, eps=eps)
    # to_out + bias + residual in the same launch, with and without the image tokens
    wo, bo = h(torch.randn(c, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(c, generator=g))
    op = k.pack_attn_out(D(wo), D(bo), heads)
    for o_attn, kw in ((out, {}), (out_ip, dict(ip_frag=frag_ip, ip_len=li, ip_scale=ip_scale))):
        full = k.cross_attn_fused(xd, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc, eps=eps, out_proj=op, **kw)
        close(full, x + o_attn.float().cpu() @ wo.T + bo, rel=1e-3, name="fused text cross-attention + to_out + residual vs fp32 torch")
        close(full, k.gemm(o_attn, D(wo), D(bo), residual=xd), rel=1e-3, name="to_out inside the launch vs the GEMM behind it")
        buf = xd.clone()
        assert torch.equal(full, k.cross_attn_fused(buf, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rpc,
                                                    eps=eps, out_proj=op, out=buf, **kw))


@pytest.mark.parametrize("n,hh,ww,c1,c2,groups,fps,silu,perm", [
    (4, 8, 8, 32, 0, 8, 1, True, False), (2, 16, 16, 320, 0, 32, 1, True, False),
    (4, 8, 8, 64, 0, 32, 4, False, True), (2, 20, 20, 64, 32, 32, 1, True, False),
    (2, 4, 4, 1280, 1280, 32, 1, True, False), (8, 5, 7, 40, 0, 4, 2, False, False),
    (2, 32, 32, 320, 0, 32, 2, False, True),
    # batches that fill the chip at the small levels run the one-launch slab kernel (whole statistics group in LDS)
    (32, 8, 8, 1280, 0, 32, 1, True, False), (32, 16, 16, 1280, 1280, 32, 1, True, False),
    (32, 32, 32, 640, 0, 32, 1, True, False), (32, 16, 16, 1920, 0, 32, 1, False, False),
    (64, 8, 8, 640, 0, 32, 4, False, True), (32, 8, 8, 1280, 640, 32, 1, True, False)])
def test_groupnorm(dev, n, hh, ww, c1, c2, groups, fps, silu, perm):
    _groupnorm_case(dev, n, hh, ww, c1, c2, groups, fps, silu, perm, mean=0.5, std=2.0)


@pytest.mark.parametrize("n,hh,cin,cout,mean", [(16, 32, 320, 320, 0.0), (4, 64, 320, 320, 3.0), (32, 32, 640, 640, 0.5), (32, 32, 320, 640, -2.0),
                                                 (32, 16, 320, 1280, 0.0)])
def test_groupnorm_statistics_from_the_conv_epilogue(dev, n, hh, cin, cout, mean):
    """conv1 -> norm2 of ResnetBlock2D (unet:203-214): the convolution's epilogue writes the GroupNorm partials of its result
    (i2v_gemm_params.gn_partial: per image, row tile and group the mean and M2 of the fp32 accumulators + bias + time-embedding
    row), the norm skips its statistics pass (i2v_gn_params.gpartial_in).  Against fp32 torch on the convolution's own fp16
    output, and against the norm's statistics pass on the same tensor.  A large per-channel offset (bias + time embedding `mean`
    standard deviations away) must not cost digits: the column constants never enter the sums of squares."""
    k = K()
    g = torch.Generator().manual_seed(n + hh + cout)
    x = h(torch.randn(n, cin, hh, hh, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g) * 0.3 + mean * 10)
    tv = h(torch.randn(2, cout, generator=g) + mean * 10)
    ga, be = h(1 + 0.3 * torch.randn(cout, generator=g)), h(0.3 * torch.randn(cout, generator=g))
    D = lambda t: t.half().to(dev)
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = _pack_conv(w).to(dev)
    rpv = (n // 2) * hh * hh
    out, st = k.conv3x3(xt, wp, D(b), rowvec=D(tv), rows_per_vec=rpv, gn_stats_groups=32)
    plain = k.conv3x3(xt, wp, D(b), rowvec=D(tv), rows_per_vec=rpv)
    assert torch.equal(out, plain), "the statistics must not change the convolution's result"
    assert st is not None, "an un-split 3x3 convolution of whole row tiles writes the partials"
    part, rows = st
    assert rows in (128, 256) and tuple(part.shape) == (n, hh * hh // rows, 32, 2) and torch.isfinite(part).all()
    y = k.groupnorm(out, D(ga), D(be), 32, 1e-6, silu=True, stats=st)
    ref = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), 32, ga, be, eps=1e-6)).permute(0, 2, 3, 1)
    close(y, ref, rel=2e-3, name="GroupNorm from the convolution's partials vs fp32 torch")
    close(y, k.groupnorm(out, D(ga), D(be), 32, 1e-6, silu=True), rel=2e-3, name="... vs the norm's own statistics pass")
    # the merged statistics themselves: mean and variance per (image, group) from the partials against torch on the fp16 output
    cpg = cout // 32
    xo = out.float().cpu().view(n, hh * hh, 32, cpg)
    mean_ref, var_ref = xo.mean(dim=(1, 3)), xo.var(dim=(1, 3), unbiased=False)
    pm, pm2 = part[..., 0].cpu().double(), part[..., 1].cpu().double()
    mean_got = pm.mean(dim=1)
    var_got = (pm2.sum(dim=1) + (rows * cpg) * ((pm - mean_got[:, None]) ** 2).sum(dim=1)) / (hh * hh * cpg)
    assert (mean_got - mean_ref).abs().max() < 2e-3 * (1 + mean_ref.abs().max()), "group means from the partials"
    assert ((var_got - var_ref).abs() / var_ref).max() < 5e-3, "group variances from the partials"
    # where the form does not exist (a convolution that splits K, a residual epilogue) the caller is told so
    small = k.conv3x3(xt[:1, :8, :8].contiguous(), wp, D(b), gn_stats_groups=32)
    assert small[1] is None and tuple(small[0].shape) == (1, 8, 8, cout)
    # ... and with an output scale: the partials would describe acc + bias, not the tensor that is stored
    scaled = k.conv3x3(xt, wp, D(b), rowvec=D(tv), rows_per_vec=rpv, gn_stats_groups=32, out_scale=0.5)
    assert scaled[1] is None and tuple(scaled[0].shape) == tuple(out.shape)
    with pytest.raises(ValueError, match="stats"):
        k.groupnorm(out, D(ga), D(be), 32, 1e-6, silu=True, stats=(part[:, :1].contiguous(), rows))


@pytest.mark.parametrize("mean,std", [(30.0, 0.5), (-200.0, 1.0), (1000.0, 4.0)])
def test_groupnorm_large_mean(dev, mean, std):
    """|mean| >> std (real SD activations have such channels): the statistics must not lose their digits to
    E[x^2] - mean^2 cancellation (per-chunk shifted sums + Chan merge).  Spatial and clip-wide statistics."""
    _groupnorm_case(dev, 4, 32, 32, 320, 0, 32, 1, True, False, mean=mean, std=std)
    _groupnorm_case(dev, 4, 16, 16, 64, 64, 32, 2, False, True, mean=mean, std=std)
    _groupnorm_case(dev, 32, 16, 16, 640, 0, 32, 1, True, False, mean=mean, std=std)      # slab kernel


@pytest.mark.parametrize("n,hh,c,n_out,fps,mean", [(32, 32, 320, 320, 1, 0.5), (32, 32, 320, 640, 16, 2.0),
                                                   (32, 64, 320, 320, 16, 0.0), (16, 32, 640, 640, 1, 20.0)])
def test_groupnorm_folded_into_gemm(dev, n, hh, c, n_out, fps, mean):
    """GroupNorm -> Linear as ONE GEMM over per-statistics-group scaled weights (i2v_groupnorm_fold_f16 +
    i2v_gemm_params.w_batch_stride / rows_per_w), and for the motion modules' entry (statistics over a clip's frames) the
    (b, frame, pixel) -> (b, pixel, frame) row gather in the same GEMM (a_perm_*)."""
    k = K()
    g = torch.Generator().manual_seed(n + c + fps)
    x = h(torch.randn(n, c, hh, hh, generator=g) * 1.5 + mean)
    ga, be = h(1 + 0.3 * torch.randn(c, generator=g)), h(0.3 * torch.randn(c, generator=g))
    w = h(torch.randn(n_out, c, generator=g) / math.sqrt(c))
    b = h(torch.randn(n_out, generator=g))
    hw = hh * hh
    if fps == 1:
        y = F.group_norm(x, 32, ga, be, eps=1e-6)
    else:   # statistics over (C / G, frames, H, W) of each clip
        xc = x.view(n // fps, fps, c, hh, hh).permute(0, 2, 1, 3, 4)
        y = F.group_norm(xc, 32, ga, be, eps=1e-6).permute(0, 2, 1, 3, 4).reshape(n, c, hh, hh)
    tok = y.permute(0, 2, 3, 1).reshape(n * hw, c)
    if fps > 1:
        tok = tok.view(n // fps, fps, hw, c).permute(0, 2, 1, 3).reshape(n * hw, c)
    ref = tok @ w.T + b
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    w_s, b_s = k.groupnorm_fold(xt, ga.half().to(dev), be.half().to(dev), 32, 1e-6, w.half().to(dev), b.half().to(dev),
                                frames_per_stat=fps)
    assert w_s.shape == (n // fps, n_out, c) and b_s.shape == (n // fps, n_out)
    kw = dict(rowvec=b_s, rows_per_vec=fps * hw, w_rows=fps * hw, a_perm=(fps, hw) if fps > 1 else None)
    assert k.gemm(xt.view(-1, c), w_s, None, query_batch_support=True, **kw)
    got = k.gemm(xt.view(-1, c), w_s, None, **kw)
    close(got, ref, name="GroupNorm folded into proj_in")


def test_gemm_batched_weights_unsupported_shapes_fail_loudly(dev):
    import i2v_adapter_unofficial_amd as pkg
    k = K()
    x = torch.randn(2048, 320, device=dev).half()
    w = torch.randn(8, 320, 320, device=dev).half()
    assert not k.gemm(x, w, None, w_rows=256, query_batch_support=True)      # too few tiles for the 8-wave kernel
    with pytest.raises(pkg.HipLibraryError, match="not implemented for this"):
        k.gemm(x, w, None, w_rows=256)


def _groupnorm_case(dev, n, hh, ww, c1, c2, groups, fps, silu, perm, mean, std):
    k = K()
    g = torch.Generator().manual_seed(n + c1)
    c = c1 + c2
    x = h(torch.randn(n, c, hh, ww, generator=g) * std + mean)
    ga, be = h(torch.randn(c, generator=g)), h(torch.randn(c, generator=g))
    if fps == 1:
        ref = F.group_norm(x, groups, ga, be, eps=1e-5)
    else:   # statistics over (C/G, fps frames, H, W)   (TransformerTemporalModel.norm)
        xr = x.view(n // fps, fps, c, hh, ww).permute(0, 2, 1, 3, 4)
        ref = F.group_norm(xr, groups, ga, be, eps=1e-5).permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
    if silu:
        ref = F.silu(ref)
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    x1, x2 = (xt, None) if c2 == 0 else (xt[..., :c1].contiguous(), xt[..., c1:].contiguous())
    out = k.groupnorm(x1, ga.half().to(dev), be.half().to(dev), groups, 1e-5, x2=x2, silu=silu, frames_per_stat=fps,
                      out_perm=perm, frames=fps)
    ref_t = ref.permute(0, 2, 3, 1)
    if perm:
        ref_t = ref_t.reshape(n // fps, fps, hh * ww, c).permute(0, 2, 1, 3).reshape(n * hh * ww, c)
    close(out, ref_t, name="groupnorm")


@pytest.mark.parametrize("M,N,K_,kind,mean", [
    (49152, 640, 320, "plain", 0.0), (33000, 1280, 128, "geglu", 1.0), (16384, 320, 640, "vt", 0.0),
    (32768, 640, 320, "plain_pe", 3.0), (16384, 320, 320, "vt_pe", 0.0), (70000, 320, 1280, "plain", 40.0)])
def test_gemm_layernorm_fold(dev, M, N, K_, kind, mean):
    """LayerNorm folded into the 8-wave GEMM's epilogue (i2v_gemm_params.ln_wsum): rstd (x W'^T - mean wsum) + W beta + b
    against LayerNorm -> Linear in fp32, every epilogue that implements it, M tails, rows with |mean| >> std."""
    from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu
    k = K()
    g = torch.Generator().manual_seed(M + N)
    x = h(torch.randn(M, K_, generator=g) * 1.5 + mean)
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    ga, be = h(1 + 0.2 * torch.randn(K_, generator=g)), h(0.2 * torch.randn(K_, generator=g))
    n = F.layer_norm(x, (K_,), ga, be, eps=1e-5)
    xd = x.half().to(dev)
    st = 1e-5            # the row statistics are computed inside the GEMM's K loop: only eps is passed
    frames = 16
    pe = h(torch.randn(32, K_, generator=g))
    if kind in ("plain", "plain_pe"):
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w, b, ga, be))
        kw = {}
        ref = n @ w.T + b
        if kind == "plain_pe":
            kw = dict(rowvec=(pe @ w.T).half().to(dev), rowvec_period=frames)
            ref = (n.view(-1, frames, K_) + pe[:frames]).view(M, K_) @ w.T + b
        assert k.gemm(xd, wf, cb, ln=(ws, st), query_ln_support=True, **kw)
        close(k.gemm(xd, wf, cb, ln=(ws, st), **kw), ref, name=f"LN-folded gemm {kind}")
    elif kind == "geglu":
        wf, ws, cb = (t.to(dev) for t in fold_layernorm_geglu(w, b, ga, be))
        y = n @ w.T + b
        close(k.gemm(xd, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, st)), y[:, : N // 2] * F.gelu(y[:, N // 2:]),
              name="LN-folded geglu")
    else:
        L = 4096 if kind == "vt" else frames
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w, b, ga, be))
        kw = {}
        nn_ = n
        if kind == "vt_pe":
            kw = dict(pe_t=(pe @ w.T).T.contiguous().half().to(dev), pe_period=frames)
            nn_ = (n.view(-1, frames, K_) + pe[:frames]).view(M, K_)
        assert k.project_vt(xd, wf, L, bias=cb, ln=(ws, st), query_ln_support=True, **kw)
        vt = k.project_vt(xd, wf, L, bias=cb, ln=(ws, st), **kw)
        ref = (nn_ @ w.T + b).view(M // L, L, N).permute(0, 2, 1)
        close(vt[:, :, :L], ref, name=f"LN-folded V^T {kind}")


_WS_CASES = [(16384, 320, "bias"), (20032, 320, "res"), (16384 + 64 * 7, 640, "ln"), (32768, 640, "ln_pe"), (16384, 960, "ln"),
             (24576, 2560, "geglu_ln"), (16384, 640, "geglu"), (65536, 320, "res_views"), (131072, 320, "res"),
             (16448, 1280, "pe")]


@pytest.mark.parametrize("M,N,kind", _WS_CASES)
def test_gemm_k320_row_major_flavours(dev, M, N, kind):
    """every epilogue flavour of the K = 320 row-major projections with >= 16384 rows (q | k | v, out-projections, GEGLU of the
    64 x 64 level) against fp32 torch, through whichever kernel the library dispatches (the 8-wave tile kernel by default)."""
    _ws_case(dev, M, N, kind)


@pytest.mark.variants
def test_gemm_weight_stationary_opt_in_child_process(dev):
    """csrc/variants/gemm_ws.hip (W slices in registers, A through three LDS stages; measured not faster, opt-in
    I2V_GEMM_WS=1 in the variants library): the same cases in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "k320_row_major"], cwd=root, env=_variants_env(I2V_GEMM_WS="1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def _ws_case(dev, M, N, kind):
    """bias, residual, LayerNorm fold (+ positional table), GEGLU (+ fold), row counts that leave 256 workgroups unequal shares,
    A / residual / C as column slices of wider matrices."""
    from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu
    k = K()
    K_ = 320
    g = torch.Generator().manual_seed(M + N)
    x = h(torch.randn(M, K_, generator=g) * 1.5 + (2.0 if "ln" in kind else 0.0))
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    ga, be = h(1 + 0.2 * torch.randn(K_, generator=g)), h(0.2 * torch.randn(K_, generator=g))
    xd, wd, bd = x.half().to(dev), w.half().to(dev), b.half().to(dev)
    frames = 16
    pe = h(torch.randn(32, K_, generator=g))
    if kind == "bias":
        close(k.gemm(xd, wd, bd, out_scale=0.5), 0.5 * (x @ w.T + b), name="ws gemm + bias")
        close(k.gemm(xd, wd), x @ w.T, name="ws gemm, no bias")
    elif kind == "res":
        r = h(torch.randn(M, N, generator=g))
        close(k.gemm(xd, wd, bd, residual=r.half().to(dev)), x @ w.T + b + r, name="ws gemm + residual")
    elif kind == "res_views":
        # A = a column slice of a wider matrix (lda 960), the residual and the output likewise (ldr, ldc = 640)
        wide = torch.zeros(M, 960, dtype=torch.float16, device=dev)
        wide[:, 320:640] = xd
        r = h(torch.randn(M, N, generator=g))
        rw = torch.zeros(M, 640, dtype=torch.float16, device=dev)
        rw[:, 320:] = r.half().to(dev)
        out = torch.full((M, 640), float("nan"), dtype=torch.float16, device=dev)
        k.gemm(wide[:, 320:640], wd, bd, residual=rw[:, 320:], out=out[:, :320])
        close(out[:, :320], x @ w.T + b + r, name="ws gemm on column-slice views")
        assert torch.isnan(out[:, 320:]).all(), "columns outside the output view were written"
    elif kind in ("ln", "ln_pe"):
        n = F.layer_norm(x, (K_,), ga, be, eps=1e-5)
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w, b, ga, be))
        kw, ref = {}, n @ w.T + b
        if kind == "ln_pe":
            kw = dict(rowvec=(pe @ w.T).half().to(dev), rowvec_period=frames)
            ref = (n.view(-1, frames, K_) + pe[:frames]).view(M, K_) @ w.T + b
        close(k.gemm(xd, wf, cb, ln=(ws, 1e-5), **kw), ref, name=f"ws gemm, LayerNorm fold {kind}")
    elif kind == "pe":
        tab = h(torch.randn(8, N, generator=g))
        close(k.gemm(xd, wd, bd, rowvec=tab.half().to(dev), rowvec_period=8), x @ w.T + b + tab.repeat(M // 8, 1),
              name="ws gemm + periodic row vector")
    elif kind == "geglu":
        from i2v_adapter_unofficial_amd.blocks import pack_geglu
        wg, bg = pack_geglu(wd, bd)
        y = x @ w.T + b
        close(k.gemm(xd, wg, bg, epilogue=k.I2V_EPI_GEGLU), y[:, : N // 2] * F.gelu(y[:, N // 2:]), name="ws geglu")
    else:
        n = F.layer_norm(x, (K_,), ga, be, eps=1e-5)
        wf, ws, cb = (t.to(dev) for t in fold_layernorm_geglu(w, b, ga, be))
        y = n @ w.T + b
        close(k.gemm(xd, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, 1e-5)), y[:, : N // 2] * F.gelu(y[:, N // 2:]),
              name="ws geglu, LayerNorm fold")


def test_gemm_layernorm_fold_unsupported_shapes_fail_loudly(dev):
    """the fold exists only in the 8-wave kernel: small problems answer `unsupported` and a forced call is an error,
    never a silently un-normalised result."""
    import i2v_adapter_unofficial_amd as pkg
    k = K()
    x = torch.randn(512, 320, device=dev).half()
    w = torch.randn(320, 320, device=dev).half()
    ws = torch.zeros(320, device=dev)
    assert not k.gemm(x, w, ln=(ws, 1e-5), query_ln_support=True)
    with pytest.raises(pkg.HipLibraryError, match="not implemented for this problem"):
        k.gemm(x, w, ln=(ws, 1e-5))
    assert not k.gemm(torch.randn(40000, 320, device=dev).half(), torch.randn(200, 320, device=dev).half(),
                      ln=(torch.zeros(200, device=dev), 1e-5), query_ln_support=True)


@pytest.mark.parametrize("rows,c,pe_period", [(100, 64, 0), (333, 320, 0), (64, 1280, 16), (40, 2560, 0), (48, 40, 8)])
def test_layernorm(dev, rows, c, pe_period):
    k = K()
    g = torch.Generator().manual_seed(rows + c)
    x = h(torch.randn(rows, c, generator=g) * 3 + 1)
    ga, be = h(torch.randn(c, generator=g)), h(torch.randn(c, generator=g))
    ref = F.layer_norm(x, (c,), ga, be, eps=1e-5)
    pe = None
    if pe_period:
        pe = h(torch.randn(32, c, generator=g))
        ref = ref + pe[:pe_period].repeat(rows // pe_period, 1)
    out = k.layernorm(x.half().to(dev), ga.half().to(dev), be.half().to(dev), 1e-5,
                      pe=None if pe is None else pe.half().to(dev), pe_period=pe_period)
    close(out, ref, name="layernorm")


@pytest.mark.parametrize("clips,frames,L,c", [(2, 4, 48, 320), (3, 2, 7, 640), (1, 16, 64, 1280)])
def test_layernorm_batched_row_blocks(dev, clips, frames, L, c):
    """the frame-0 rows of every clip (i2v:484) normalised IN PLACE from the [clips * frames * L, C] token matrix (a 3-D
    view with the clip as batch stride): bit-identical to LayerNorm of the gathered copy, and to torch."""
    k = K()
    g = torch.Generator().manual_seed(clips * 100 + L)
    x = h(torch.randn(clips * frames * L, c, generator=g) * 2 + 0.5)
    ga, be = h(torch.randn(c, generator=g)), h(torch.randn(c, generator=g))
    xd = x.half().to(dev)
    view = xd.view(clips, frames * L, c)[:, :L]
    out = k.layernorm(view, ga.half().to(dev), be.half().to(dev), 1e-5)
    assert out.shape == (clips * L, c)
    gathered = view.contiguous().view(-1, c)
    assert torch.equal(out, k.layernorm(gathered, ga.half().to(dev), be.half().to(dev), 1e-5))
    close(out, F.layer_norm(x.view(clips, frames * L, c)[:, :L].reshape(-1, c), (c,), ga, be, eps=1e-5),
          name="layernorm over batched row blocks")
    with pytest.raises(ValueError):       # a batch stride that is not a multiple of 8 elements
        k.layernorm(torch.zeros(2, 9, c + 4, dtype=torch.float16, device=dev)[:, :4, :c], ga.half().to(dev),
                    be.half().to(dev), 1e-5)


def test_select_row_follows_the_device_step_counter(dev):
    """K.select_row: the row of a per-timestep table at the DEVICE-side index (clamped to the table), as the replayed step
    reads its time-embedding projections."""
    k = K()
    table = torch.randn(5, 64, generator=torch.Generator().manual_seed(2)).half().to(dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    for i, want in ((0, 0), (3, 3), (4, 4), (9, 4), (-2, 0)):
        idx.fill_(i)
        assert torch.equal(k.select_row(table, idx), table[want: want + 1])
    wide = torch.randn(3, 96, generator=torch.Generator().manual_seed(3)).half().to(dev)
    idx.fill_(1)
    assert torch.equal(k.select_row(wide[:, :64], idx), wide[1:2, :64])        # a column slice: ld > cols


def test_layout_edges_and_misc(dev):
    k = K()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 4, 6, 5, generator=g)
    t = k.nchw_to_tokens(x.to(dev), c_pad=8)
    assert t.shape == (3, 6, 5, 8)
    close(t[..., :4].permute(0, 3, 1, 2), h(x), rel=1e-6, name="nchw_to_tokens")
    assert (t[..., 4:] == 0).all()
    back = k.tokens_to_nchw(t, c=4, dtype=torch.float32)
    close(back, h(x), rel=1e-6, name="tokens_to_nchw")
    t16 = k.nchw_to_tokens(x.half().to(dev))
    close(k.tokens_to_nchw(t16), h(x), rel=1e-6, name="fp16 edge roundtrip")
    y = h(torch.randn(37, 24, generator=g))
    close(k.silu(y.half().to(dev)), F.silu(y), rel=2e-3, name="silu")
    close(k.repeat_rows(y.half().to(dev), 3), y.repeat_interleave(3, 0), rel=1e-6, name="repeat_rows")
    src = y.half().to(dev).view(1, 37, 24).expand(1, 37, 24)
    big = h(torch.randn(4, 10, 24, generator=g)).half().to(dev)
    dst = torch.zeros(4, 3, 8, dtype=torch.float16, device=dev)
    k.copy3d(big[:, 2:5, 8:16], dst)
    close(dst, big[:, 2:5, 8:16], rel=1e-6, name="copy3d")
    dst7 = torch.zeros(4, 3, 7, dtype=torch.float16, device=dev)            # odd row length / unaligned base: scalar form
    k.copy3d(big[:, 1:4, 3:10], dst7)
    close(dst7, big[:, 1:4, 3:10], rel=1e-6, name="copy3d scalar")
    wide = h(torch.randn(3, 16 * 50, 320, generator=g)).half().to(dev)      # frame-0 gather of the adapter block: 16-byte form
    first = torch.zeros(3, 50, 320, dtype=torch.float16, device=dev)
    k.copy3d(wide[:, :50], first)
    assert torch.equal(first, wide[:, :50])


def test_timestep_embedding(dev):
    from oracle.blocks import Timesteps
    k = K()
    t = torch.tensor([0.0, 1.0, 40.0, 500.0, 999.0])
    ref = Timesteps(320, True, 0)(t)
    out = k.timestep_embedding(t.to(dev), 320)
    close(out, ref, rel=2e-3, name="timestep embedding")
    idx = torch.tensor([3], dtype=torch.int32, device=dev)
    out = k.timestep_embedding(t.to(dev), 320, t_index=idx)
    close(out, ref[3:4], rel=2e-3, name="timestep embedding (indexed)")
    # a step counter past the table (a graph replayed more often than the schedule is long) is clamped, never read past
    idx.fill_(17)
    close(k.timestep_embedding(t.to(dev), 320, t_index=idx), ref[4:5], rel=2e-3, name="timestep embedding (clamped)")


def test_ddim_prep_and_step(dev):
    from oracle.blocks import DDIMScheduler
    k = K()
    g = torch.Generator().manual_seed(4)
    b, f, c, hh, ww = 2, 3, 4, 6, 5
    lat = torch.randn(b, f, c, hh, ww, generator=g)
    cond = torch.randn(b, c, hh, ww, generator=g)
    lat_d = lat.clone().to(dev)
    mi = k.ddim_prep(lat_d, cond.to(dev), 8, 2)
    ref_lat = lat.clone()
    ref_lat[:, 0] = cond
    assert torch.equal(lat_d.cpu(), ref_lat)
    assert mi.shape == (2 * b * f, hh, ww, 8)
    tok = ref_lat.reshape(b * f, c, hh, ww).permute(0, 2, 3, 1)
    close(mi[: b * f, ..., :4], h(tok), rel=1e-6, name="prep copy 0")
    close(mi[b * f:, ..., :4], h(tok), rel=1e-6, name="prep copy 1")
    sch = DDIMScheduler()
    sch.set_timesteps(25)
    ts = sch.timesteps[3:]
    coef = []
    for t in ts.tolist():
        pt = t - 1000 // 25
        a_t = sch.alphas_cumprod[t]
        a_p = sch.alphas_cumprod[pt] if pt >= 0 else sch.final_alpha_cumprod
        coef.append([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()])
    coef = torch.tensor(coef, dtype=torch.float32)
    npred = h(torch.randn(2 * b * f, hh, ww, 4, generator=g))
    step = torch.tensor([5], dtype=torch.int32, device=dev)
    k.ddim_cfg_step(lat_d, npred.half().to(dev), coef.to(dev), step, 7.5, 2)
    assert int(step.item()) == 6
    u, cnd = npred[: b * f], npred[b * f:]
    eps = (u + 7.5 * (cnd - u)).permute(0, 3, 1, 2).reshape(b, f, c, hh, ww)
    ref = sch.step(eps, ts[5], ref_lat)
    close(lat_d, ref, rel=1e-5, name="ddim step")
    # the device-side counter wraps at the end of the coefficient table and a stale index is clamped
    n = coef.shape[0]
    step.fill_(n - 1)
    lat2 = ref_lat.clone().to(dev)
    k.ddim_cfg_step(lat2, npred.half().to(dev), coef.to(dev), step, 7.5, 2)
    assert int(step.item()) == 0
    close(lat2, sch.step(eps, ts[n - 1], ref_lat), rel=1e-5, name="ddim last step")
    step.fill_(n + 40)
    lat3 = ref_lat.clone().to(dev)
    k.ddim_cfg_step(lat3, npred.half().to(dev), coef.to(dev), step, 7.5, 2)
    assert int(step.item()) == 0 and torch.equal(lat3, lat2)
    # fp32 noise prediction (the UNet's conv_out result un-rounded): exact against the host formula on fp32 values
    np32 = torch.randn(2 * b * f, hh, ww, 4, generator=g)
    step.fill_(5)
    lat4 = ref_lat.clone().to(dev)
    k.ddim_cfg_step(lat4, np32.to(dev), coef.to(dev), step, 7.5, 2)
    u32, c32 = np32[: b * f], np32[b * f:]
    eps32 = (u32 + 7.5 * (c32 - u32)).permute(0, 3, 1, 2).reshape(b, f, c, hh, ww)
    close(lat4, sch.step(eps32, ts[5], ref_lat), rel=1e-6, name="ddim step, fp32 noise prediction")


@pytest.mark.parametrize("b,f,c,hh,ww,sigma", [(2, 3, 4, 8, 6, 1.0), (1, 16, 4, 64, 64, 0.37), (2, 2, 4, 1, 5, 1.9)])
def test_first_frame_prior(dev, b, f, c, hh, ww, sigma):
    """pipe:647-656 in one kernel vs the oracle's torch ops (torchvision-style 3x3 Gaussian, reflect padding)."""
    from oracle.blocks import DDIMScheduler, gaussian_blur3
    k = K()
    g = torch.Generator().manual_seed(b + f + hh)
    cond = torch.randn(b, c, hh, ww, generator=g)
    u = torch.rand(b, f, c, hh, ww, generator=g)
    noise = torch.randn(b, f, c, hh, ww, generator=g)
    sch = DDIMScheduler()
    sch.set_timesteps(25)
    t = sch.timesteps[2]
    if hh > 1:
        blurred = gaussian_blur3(cond, sigma)
    else:   # a 1-pixel axis cannot be reflect-padded by torch; the kernel degenerates to the centre row
        xs = torch.linspace(-1.0, 1.0, 3)
        k1 = torch.exp(-0.5 * (xs / sigma) ** 2)
        k1 = k1 / k1.sum()
        xp = F.pad(cond, (1, 1, 0, 0), mode="reflect")
        blurred = (k1[0] * xp[..., :-2] + k1[1] * xp[..., 1:-1] + k1[2] * xp[..., 2:]) * (k1[1] + 2 * k1[0])
    mask = (u < 0.6).float()
    prior = mask * blurred.unsqueeze(1) + (1 - mask) * cond.unsqueeze(1)
    ref = sch.add_noise(prior, noise, t.repeat(b))
    a = float(sch.alphas_cumprod[int(t)])
    out = k.first_frame_prior(cond.to(dev), u.to(dev), noise.to(dev), sigma, 0.6, a ** 0.5, (1 - a) ** 0.5)
    close(out, ref, rel=2e-6, name="first-frame prior + add_noise")


def test_bad_arguments_raise(dev):
    import i2v_adapter_unofficial_amd as pkg
    k = K()
    with pytest.raises(pkg.HipLibraryError):
        k.gemm(torch.zeros(4, 8, dtype=torch.float16), torch.zeros(4, 8, dtype=torch.float16))     # CPU tensors
    with pytest.raises(pkg.HipLibraryError):
        k.gemm(torch.zeros(4, 12, dtype=torch.float16, device=dev), torch.zeros(4, 12, dtype=torch.float16, device=dev))
    with pytest.raises(pkg.HipLibraryError):
        k.attention(torch.zeros(8, 12, dtype=torch.float16, device=dev), torch.zeros(8, 12, dtype=torch.float16, device=dev),
                    torch.zeros(1, 12, 8, dtype=torch.float16, device=dev), batch_q=1, lq=8, lk=8, heads=1, head_dim=12)


@pytest.mark.variants
def test_gemm_4wave_two_workgroups_per_cu_variant(dev):
    """the 4-wave form of the big GEMM kernel (128 x 320 x 32 tiles, two workgroups per CU; gemm_big.hip `NW`) measured
    slower on every shape of the step and is off by default; with I2V_GEMM_4W=1 (read once per process) every eligible
    plain GEMM takes it.  The GEMM / LayerNorm-fold / GroupNorm-fold / V^T test set is re-run that way in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "(gemm or fold or project_vt) and not 4wave and not conv"], cwd=root,
                       env=_variants_env(I2V_GEMM_4W="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.variants
def test_gemm_alternating_groups_opt_in_child_process(dev):
    """I2V_GEMM_ALT=2 (read once per process) sends the short-K GEMM flavours to the alternating-groups kernel (csrc/variants/gemm_alt.hip:
    one wave group in the K loop of a 128-row tile while the other runs the previous tile's epilogue; measured slower, kept
    as a switch): tools/alt_ab.py checks every flavour against a torch fp32 reference in a child process."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "alt_ab.py")], cwd=root, env=_variants_env(I2V_GEMM_ALT="2"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    errs = [float(m) for m in re.findall(r"rel err ([0-9.e+-]+)", r.stdout)]
    assert len(errs) == 11 and max(errs) < 2e-3 and "finite False" not in r.stdout, r.stdout

"""Tensor-level wrappers over the C ABI (include/i2v_hip.h).

PyTorch is used only for device memory and the stream: every function validates its operands on the host
(shape / dtype / contiguity, so a kernel never sees shapes its grid does not assume), allocates the output with
torch.empty and launches the HIP kernel on torch's current stream through ctypes.  Nothing here computes with
torch ops and nothing falls back to the CPU: a CPU tensor or a missing library raises.
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (I2V_A_CONV3X3, I2V_A_PLAIN, I2V_EPI_GEGLU, I2V_EPI_GELU, I2V_EPI_NONE, I2V_STORE_ROWMAJOR,
                   I2V_STORE_ROWPERM, I2V_STORE_VT, I2V_STORE_VT_T, AttnBwdParams, AttnParams, GemmParams, GnParams,
                   HipLibraryError, LnParams, TAttnParams)

f16 = torch.float16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, name: str, dtype=f16):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise HipLibraryError(f"{name} is on {t.device}: the I2V-Adapter HIP path has no CPU fallback "
                              "(move tensors to a ROCm device)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _mat(t: torch.Tensor, name: str):
    """2-D fp16 matrix whose rows are unit-stride; returns (tensor, leading dimension)."""
    _req(t, name)
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name} must have unit stride along its last dim")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return t, ld


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


# ---------------------------------------------------------------------------------------------- the precise residual stream
# (i2v_gemm_params.residual_lo / c_lo)  The stream between the UNet's modules as an fp16 pair: the tensor every kernel
# reads as before (hi) carries its low half -- the bits its fp16 rounding dropped -- as an attribute.  A producer called with
# `precise=True` adds the residual's low half (when the residual has one) and attaches the result's.
def lo_of(t):
    """the low half that travels with a stream tensor, or None"""
    return None if t is None else getattr(t, "_i2v_lo", None)


def sview(t, *shape):
    """t.view(*shape) that keeps the low half attached"""
    v, lo = t.view(*shape), lo_of(t)
    if lo is not None:
        v._i2v_lo = lo.view(*shape)
    return v


def _attach_lo(p_res_lo_setter, residual, out):
    """(residual's low half or None after checking it is laid out like the residual, fresh low half for `out`)"""
    rl = lo_of(residual)
    if rl is not None and (rl.shape != residual.shape or rl.stride() != residual.stride() or rl.dtype != f16):
        raise ValueError("the low half of a stream tensor must be laid out like the tensor")
    cl = torch.empty_like(out)
    if cl.stride() != out.stride():
        raise ValueError("precise stream: `out` must be dense")
    p_res_lo_setter(rl, cl)
    out._i2v_lo = cl
    return rl, cl


# ---------------------------------------------------------------------------------------------- GEMM / conv
def gemm(a, w, bias=None, *, a2=None, residual=None, rowvec=None, rows_per_vec=0, epilogue=I2V_EPI_NONE,
         out=None, store=I2V_STORE_ROWMAJOR, frames=0, hw=0, vt_len=0, vt_ld=0, out_scale=1.0, ln=None,
         rowvec_period=0, query_ln_support=False, w_rows=0, a_perm=None, query_batch_support=False, precise=False):
    """C = epi(A W^T + bias + rowvec + residual) * out_scale   (see i2v_gemm_f16).

    precise: the result is a tensor of the precise residual stream -- its low half is written beside it (`lo_of(out)`), and the
    residual's low half, if it has one, is added with it (row-major / row-permuted plain stores only).

    w may be a stack [S, N, K] of weight matrices with w_rows = rows of A per matrix (GroupNorm folded into proj_in,
    groupnorm_fold).  a_perm = (frames, hw): A's rows are (batch, frame, pixel) and are read as (batch, pixel, frame).
    query_batch_support=True launches nothing and returns whether the library implements those two for this problem.

    ln = (wsum fp32 [N], eps): LayerNorm of A's rows folded into the GEMM (w = W o gamma, bias = W beta + b; the row
    statistics are computed inside the kernel; see i2v_gemm_params.ln_wsum).  rowvec_period > 0: rowvec row = m % period (with a VT_T store the
    table is passed transposed, [N, >= period]).  query_ln_support=True launches nothing and returns whether the library
    implements the fold for exactly this problem."""
    lib = _lib.load()
    a, lda = _mat(a, "a")
    w_stack = None
    if w.dim() == 3:
        _req(w, "w")
        if w_rows <= 0 or not w.is_contiguous() or a.shape[0] != w.shape[0] * w_rows:
            raise ValueError(f"stacked w {tuple(w.shape)} needs contiguous storage and w_rows with S * w_rows == M")
        w_stack, w = w, w[0]
    w, ldw = _mat(w, "w")
    M, K1 = a.shape
    N, K = w.shape
    p = GemmParams()
    if w_stack is not None:
        p.w_batch_stride, p.rows_per_w = w_stack.stride(0), w_rows
    if a_perm is not None:
        p.a_perm_frames, p.a_perm_hw = int(a_perm[0]), int(a_perm[1])
    p.a, p.lda = _p(a), lda
    if a2 is not None:
        a2, lda2 = _mat(a2, "a2")
        if a2.shape[0] != M or K1 + a2.shape[1] != K:
            raise ValueError(f"dual-source A shapes {tuple(a.shape)} + {tuple(a2.shape)} do not match W {tuple(w.shape)}")
        p.a2, p.lda2, p.k_split = _p(a2), lda2, K1
    elif K1 != K:
        raise ValueError(f"A is {tuple(a.shape)} but W is {tuple(w.shape)}")
    p.a_mode = I2V_A_PLAIN
    p.w, p.ldw = _p(w), ldw
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous vector of N elements")
        p.bias = _p(bias)
    n_out = N // 2 if epilogue == I2V_EPI_GEGLU else N
    if store in (I2V_STORE_VT, I2V_STORE_VT_T):
        tokens, chans = (N, M) if store == I2V_STORE_VT else (M, N)
        if out is None or vt_len <= 0 or vt_ld < vt_len or tokens % vt_len != 0:
            raise ValueError("VT store needs an `out` buffer, vt_len > 0, vt_ld >= vt_len and tokens % vt_len == 0")
        _req(out, "out")
        if not out.is_contiguous() or out.numel() < (tokens // vt_len) * chans * vt_ld:
            raise ValueError("VT `out` must be contiguous with at least (tokens / vt_len) * channels * vt_ld elements")
        p.c, p.ldc = _p(out), vt_ld
        p.vt_len, p.vt_ld = vt_len, vt_ld
    else:
        if out is None:
            out = torch.empty((M, n_out), dtype=f16, device=a.device)
        out, ldc = _mat(out, "out")
        if out.shape != (M, n_out):
            raise ValueError(f"out must be {(M, n_out)}, got {tuple(out.shape)}")
        p.c, p.ldc = _p(out), ldc
    if residual is not None:
        residual, ldr = _mat(residual, "residual")
        if residual.shape != (M, n_out) or epilogue == I2V_EPI_GEGLU:
            raise ValueError(f"residual must be {(M, n_out)} (and is not supported with GEGLU)")
        p.residual, p.ldr = _p(residual), ldr
    if rowvec is not None:
        rowvec, ldv = _mat(rowvec, "rowvec")
        if rowvec_period > 0:
            want = (N, rowvec_period) if store == I2V_STORE_VT_T else (rowvec_period, N)
            if rowvec.shape[0] < want[0] or rowvec.shape[1] < want[1] or (store != I2V_STORE_VT_T and rowvec.shape[1] != N):
                raise ValueError(f"periodic rowvec must cover {want}, got {tuple(rowvec.shape)}")
            p.rowvec, p.ld_rowvec, p.rowvec_period = _p(rowvec), ldv, rowvec_period
        else:
            if rows_per_vec <= 0 or M % rows_per_vec != 0 or rowvec.shape != (M // rows_per_vec, N):
                raise ValueError(f"rowvec must be [M / rows_per_vec, N], got {tuple(rowvec.shape)}")
            p.rowvec, p.ld_rowvec, p.rows_per_vec = _p(rowvec), ldv, rows_per_vec
    if ln is not None:
        wsum, eps = ln
        _req(wsum, "ln wsum", dtype=torch.float32)
        if wsum.numel() != N or not wsum.is_contiguous() or a2 is not None:
            raise ValueError(f"ln = (wsum fp32 [N], eps) with a single-source A expected, got {tuple(wsum.shape)}")
        p.ln_wsum, p.ln_eps = _p(wsum), float(eps)
    p.M, p.N, p.K = M, N, K
    p.epilogue, p.store_mode = epilogue, store
    p.frames, p.hw = frames, hw
    p.out_scale = out_scale
    if query_ln_support:
        return bool(lib.i2v_gemm_ln_supported(C.byref(p)))
    if query_batch_support:
        return bool(lib.i2v_gemm_batch_supported(C.byref(p)))
    keep = None
    if precise:
        if epilogue != I2V_EPI_NONE or store not in (I2V_STORE_ROWMAJOR, I2V_STORE_ROWPERM) or ln is not None:
            raise ValueError("precise=True needs a plain row-major / row-permuted store")

        def set_lo(rl, cl):
            p.residual_lo, p.c_lo = _p(rl), _p(cl)
        keep = _attach_lo(set_lo, residual, out)
    ws = _attach_splitk_workspace(lib, p, a.device)
    _lib.check(lib.i2v_gemm_f16(C.byref(p), _stream()), "i2v_gemm_f16")
    del ws, keep
    return out


def _attach_splitk_workspace(lib, p, device):
    """Small-M / long-K problems split K over workgroups; the fp32 partial tiles live in a caller-owned scratch."""
    need = lib.i2v_gemm_workspace_bytes(C.byref(p))
    if need <= 0:
        return None
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    p.workspace, p.workspace_bytes = _p(ws), need
    return ws


def conv_k_block(cin: int) -> int:
    """contraction order of the 3x3 convolution for this (padded) channel count: 64 = channel-block-major (the 9 taps
    of a 64-channel block are consecutive, i2v_gemm_params.conv_kblock), 0 = tap-major.  `blocks.pack_conv3x3` lays the
    weights out by the same rule."""
    return 64 if cin % 64 == 0 and os.environ.get("I2V_CONV_KBLOCK", "1") != "0" else 0


def conv3x3(x, w_packed, bias=None, *, stride=1, upsample=False, rowvec=None, rows_per_vec=0, residual=None,
            out_scale=1.0, asym_pad=False, out_f32=False, gn_stats_groups=0, precise=False, output_size=None):
    """3x3 / pad 1 convolution of a token-major image x [N, H, W, Cin] with w_packed [Cout, 9 * Cin]
    (k ordered as `blocks.pack_conv3x3` lays it out: tap-major, or channel-block-major when Cin % 64 == 0, see
    conv_k_block); optional nearest-2x upsampling of the input first.  asym_pad (stride 2): no
    padding at the top / left, one zero row / column at the bottom / right (the VAE encoder's Downsample2D(padding=0)).
    out_f32: the result stays fp32 (narrow outputs only, Cout <= 64: the UNet's conv_out feeding the DDIM / CFG kernel).
    gn_stats_groups > 0: returns (out, stats) -- stats = the GroupNorm statistics of `out` over that many channel groups as
    partials written by the convolution's epilogue (i2v_gemm_params.gn_partial), to hand to `groupnorm(out, ..., stats=stats)`,
    or None where the epilogue form is not implemented for this problem (the norm then runs its own statistics pass).
    precise: as `gemm` (the result and the residual are tensors of the precise residual stream).
    output_size (upsample only): (2 H, 2 W) or one less in either dimension -- `F.interpolate(size=output_size, mode="nearest")`
    in front of the convolution, the reference's forward_upsample_size path for latent sizes that are not multiples of 8
    (unet:1304-1311, 1414-1415)."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"x must be a contiguous [N, H, W, C] tensor, got {tuple(x.shape)}")
    n, h, wd, cin = x.shape
    w_packed, ldw = _mat(w_packed, "w_packed")
    cout, K = w_packed.shape
    if K != 9 * cin:
        raise ValueError(f"w_packed is {tuple(w_packed.shape)} but x has {cin} channels")
    if asym_pad and (stride != 2 or upsample):
        raise ValueError("asym_pad needs stride 2 and no upsampling")
    if upsample:
        if stride != 1:
            raise ValueError("upsample conv must have stride 1")
        oh, ow = 2 * h, 2 * wd
        if output_size is not None:
            oh, ow = int(output_size[0]), int(output_size[1])
            if oh not in (2 * h, 2 * h - 1) or ow not in (2 * wd, 2 * wd - 1):
                raise NotImplementedError(f"upsample to {(oh, ow)} from {(h, wd)}: only 2x and 2x - 1 (a level that was odd before its "
                                          "stride-2 down-sampler) occur on the UNet's path")
    else:
        if output_size is not None:
            raise ValueError("output_size belongs to an upsampling convolution")
        padsum = 1 if asym_pad else 2
        oh, ow = (h + padsum - 3) // stride + 1, (wd + padsum - 3) // stride + 1
    M = n * oh * ow
    if out_f32 and (cout > 64 or residual is not None):
        raise ValueError("out_f32 is for narrow outputs (Cout <= 64) without a residual")
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32 if out_f32 else f16, device=x.device)
    p = GemmParams()
    p.c_is_f32 = 1 if out_f32 else 0
    p.a, p.lda = _p(x), cin
    p.a_mode = I2V_A_CONV3X3
    p.w, p.ldw = _p(w_packed), ldw
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != cout or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous vector of Cout elements")
        p.bias = _p(bias)
    if residual is not None:
        _req(residual, "residual")
        if tuple(residual.shape) != (n, oh, ow, cout) or not residual.is_contiguous():
            raise ValueError(f"residual must be contiguous {(n, oh, ow, cout)}")
        p.residual, p.ldr = _p(residual), cout
    if rowvec is not None:
        rowvec, ldv = _mat(rowvec, "rowvec")
        if rows_per_vec <= 0 or M % rows_per_vec != 0 or rowvec.shape != (M // rows_per_vec, cout):
            raise ValueError(f"rowvec must be [M / rows_per_vec, Cout], got {tuple(rowvec.shape)}")
        p.rowvec, p.ld_rowvec, p.rows_per_vec = _p(rowvec), ldv, rows_per_vec
    p.c, p.ldc = _p(out), cout
    p.M, p.N, p.K = M, cout, K
    p.epilogue, p.store_mode = I2V_EPI_NONE, I2V_STORE_ROWMAJOR
    p.out_scale = out_scale
    p.n_img, p.in_h, p.in_w, p.cin = n, h, wd, cin
    p.out_h, p.out_w, p.stride, p.upsample = oh, ow, stride, 1 if upsample else 0
    p.asym_pad = 1 if asym_pad else 0
    p.conv_kblock = conv_k_block(cin)
    keep = None
    if precise:
        if out_f32 or gn_stats_groups:
            raise ValueError("precise=True is an fp16 result without GroupNorm partials")

        def set_lo(rl, cl):
            p.residual_lo, p.c_lo = _p(rl), _p(cl)
        keep = _attach_lo(set_lo, residual, out)
    ws = _attach_splitk_workspace(lib, p, x.device)
    stats = None
    if gn_stats_groups:
        p.gn_groups = int(gn_stats_groups)
        # (asked with the split-K workspace attached: a problem that splits K keeps doing so -- its partial tiles have no epilogue
        #  of their own -- and the norm runs its statistics pass)
        rows = lib.i2v_gemm_gn_partial_rows(C.byref(p))
        if rows > 0:
            part = torch.empty((n, (oh * ow) // rows, int(gn_stats_groups), 2), dtype=torch.float32, device=x.device)
            p.gn_partial = _p(part)
            stats = (part, rows)
    _lib.check(lib.i2v_gemm_f16(C.byref(p), _stream()), "i2v_gemm_f16(conv3x3)")
    del ws, keep
    return (out, stats) if gn_stats_groups else out


def project_vt(tokens, w_v, batch_len, out=None, bias=None, ln=None, pe_t=None, pe_period=0, query_ln_support=False):
    """V^T[batch][channel][key] = (tokens W_v^T)^T, emitted directly by the GEMM epilogue (I2V_STORE_VT).
    tokens [batches * batch_len, K]; w_v [C, K]; returns [batches, C, pad8(batch_len)] (tail keys unwritten).
    ln / bias / pe_t: LayerNorm(+positional table, transposed [C, >= pe_period]) folded into the projection (see gemm)."""
    tokens, _ = _mat(tokens, "tokens")
    T = tokens.shape[0]
    if T % batch_len != 0:
        raise ValueError(f"{T} tokens are not a multiple of batch_len {batch_len}")
    ld = pad8(batch_len)
    Cc = w_v.shape[0]
    natural = Cc % 320 == 0 and batch_len % 4 == 0 and T >= 8192
    if out is None:
        out = torch.empty((T // batch_len, Cc, ld), dtype=f16, device=tokens.device)
    if query_ln_support:
        return natural and gemm(tokens, w_v, bias, store=I2V_STORE_VT_T, vt_len=batch_len, vt_ld=ld, ln=ln, rowvec=pe_t,
                                rowvec_period=pe_period, out=out, query_ln_support=True)
    if natural:
        # natural operand order (A = tokens): eligible for the 256-row LDS-DMA tile kernel, which transposes in its
        # MFMA operand order (I2V_STORE_VT_T)
        gemm(tokens, w_v, bias, store=I2V_STORE_VT_T, vt_len=batch_len, vt_ld=ld, out=out, ln=ln, rowvec=pe_t,
             rowvec_period=pe_period)
    else:
        if ln is not None or pe_t is not None or bias is not None:
            raise ValueError("the LayerNorm-folded V^T projection needs the natural operand order "
                             "(C % 320 == 0, batch_len % 4 == 0, >= 8192 tokens)")
        gemm(w_v, tokens, store=I2V_STORE_VT, vt_len=batch_len, vt_ld=ld, out=out)
    return out



# ---------------------------------------------------------------------------------------------- attention
def attention(q, k, vt, *, batch_q, lq, lk, heads, head_dim, kv_group=1, scale=None, out=None, accumulate=False,
              acc_scale=1.0, return_lse=False):
    """Flash attention forward.  q [batch_q * lq, >= heads*head_dim] (row-strided view is fine),
    k [batch_kv * lk, ...], vt [batch_kv, heads*head_dim, >= pad8(lk)], returns [batch_q * lq, heads*head_dim]
    (return_lse: the pair (out, fp32 [batch_q, heads, lq] log2-sum-exp of the scaled logits), written by the same pass: what
    attention_bwd otherwise recomputes)."""
    lib = _lib.load()
    q, ldq = _mat(q, "q")
    k, ldk = _mat(k, "k")
    _req(vt, "vt")
    Cc = heads * head_dim
    bkv = batch_q // kv_group
    if q.shape[0] != batch_q * lq or q.shape[1] < Cc:
        raise ValueError(f"q is {tuple(q.shape)}, expected [{batch_q * lq}, >={Cc}]")
    if k.shape[0] != bkv * lk or k.shape[1] < Cc:
        raise ValueError(f"k is {tuple(k.shape)}, expected [{bkv * lk}, >={Cc}]")
    if vt.dim() != 3 or not vt.is_contiguous() or vt.shape[0] != bkv or vt.shape[1] != Cc or vt.shape[2] < pad8(lk):
        raise ValueError(f"vt is {tuple(vt.shape)}, expected contiguous [{bkv}, {Cc}, >={pad8(lk)}]")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs the `out` tensor holding the previous result")
        out = torch.empty((batch_q * lq, Cc), dtype=f16, device=q.device)
    out, ldo = _mat(out, "out")
    if out.shape[0] != batch_q * lq or out.shape[1] < Cc:
        raise ValueError("bad `out` shape")
    p = AttnParams()
    p.q, p.q_row_stride, p.q_batch_stride = _p(q), ldq, lq * ldq
    p.k, p.k_row_stride, p.k_batch_stride = _p(k), ldk, lk * ldk
    p.vt, p.vt_row_stride, p.vt_batch_stride = _p(vt), vt.shape[2], Cc * vt.shape[2]
    p.o, p.o_row_stride, p.o_batch_stride = _p(out), ldo, lq * ldo
    p.batch_q, p.kv_group, p.heads, p.head_dim, p.lq, p.lk = batch_q, kv_group, heads, head_dim, lq, lk
    p.scale = float(head_dim) ** -0.5 if scale is None else scale
    p.accumulate, p.acc_scale = 1 if accumulate else 0, acc_scale
    lse = None
    if return_lse:
        if accumulate:
            raise ValueError("return_lse is not combined with accumulate")
        lse = torch.empty((batch_q, heads, lq), dtype=torch.float32, device=q.device)
        p.lse = _p(lse)
    _lib.check(lib.i2v_attention_f16(C.byref(p), _stream()), "i2v_attention_f16")
    return (out, lse) if return_lse else out


def temporal_attention(q, k, vt, *, n_pixels, frames, heads, head_dim, scale=None):
    """Motion-module attention over the frame axis; tokens in (b, pixel, frame) order.
    q, k [n_pixels * frames, >= C]; vt [n_pixels, C, >= pad8(frames)]."""
    lib = _lib.load()
    q, ldq = _mat(q, "q")
    k, ldk = _mat(k, "k")
    _req(vt, "vt")
    Cc = heads * head_dim
    if q.shape[0] != n_pixels * frames or k.shape[0] != n_pixels * frames or q.shape[1] < Cc or k.shape[1] < Cc:
        raise ValueError("q / k must be [n_pixels * frames, >= C]")
    if vt.dim() != 3 or not vt.is_contiguous() or vt.shape[0] != n_pixels or vt.shape[1] != Cc or \
            vt.shape[2] < pad8(frames):
        raise ValueError(f"vt is {tuple(vt.shape)}, expected contiguous [{n_pixels}, {Cc}, >={pad8(frames)}]")
    out = torch.empty((n_pixels * frames, Cc), dtype=f16, device=q.device)
    p = TAttnParams()
    p.q, p.q_row_stride = _p(q), ldq
    p.k, p.k_row_stride = _p(k), ldk
    p.vt, p.vt_ld = _p(vt), vt.shape[2]
    p.o, p.o_row_stride = _p(out), Cc
    p.n_pixels, p.frames, p.heads, p.head_dim = n_pixels, frames, heads, head_dim
    p.scale = float(head_dim) ** -0.5 if scale is None else scale
    _lib.check(lib.i2v_temporal_attention_f16(C.byref(p), _stream()), "i2v_temporal_attention_f16")
    return out


def motion_attn_supported(rows, channels, heads, head_dim, frames):
    """is the fused LayerNorm + q / k / v + temporal attention kernel implemented for this shape?"""
    return bool(_lib.load().i2v_motion_attn_supported(rows, channels, heads, head_dim, frames))


def _pack_head_fragments(weights, heads):
    """per head the rows of each weight, zero-padded to a multiple of 16 rows, in MFMA-fragment order
    [heads][parts][C / 32][pad16(d) / 16][lane = 16 (k chunk) + row][8] (returned as [rows, C])."""
    c_out, c_in = weights[0].shape
    d = c_out // heads
    dp = (d + 15) // 16 * 16
    n = len(weights)
    w = torch.zeros((heads, n, dp, c_in), dtype=f16, device=weights[0].device)
    for i, t in enumerate(weights):
        w[:, i, :d] = t.detach().to(f16).view(heads, d, c_in)
    w = w.view(heads, n, dp // 16, 16, c_in // 32, 4, 8).permute(0, 1, 4, 2, 5, 3, 6)      # [h, part, s, t, g, l15, 8]
    return w.contiguous().view(heads * n * dp, c_in)


def pack_motion_qkv(wq, wk, wv, heads):
    """`w_qkv` of i2v_motion_attn_f16: per head its rows of Wq, Wk, Wv in fragment order (`_pack_head_fragments`)."""
    w = _pack_head_fragments((wq, wk, wv), heads)
    if w.shape[0] != _lib.load().i2v_motion_attn_pack_rows(heads, wq.shape[0] // heads):
        raise RuntimeError("pack_motion_qkv: row count differs from i2v_motion_attn_pack_rows")
    return w


def pack_cross_q(wq, heads):
    """`w_q` of i2v_cross_attn_fused_f16: per head its rows of to_q in fragment order."""
    w = _pack_head_fragments((wq,), heads)
    if w.shape[0] != _lib.load().i2v_cross_attn_fused_pack_rows(heads, wq.shape[0] // heads):
        raise RuntimeError("pack_cross_q: row count differs from i2v_cross_attn_fused_pack_rows")
    return w


def cross_attn_fused_supported(rows, channels, heads, head_dim, ctx_len, rows_per_ctx):
    """is the fused LayerNorm + to_q + cross-attention kernel implemented for this shape?"""
    return bool(_lib.load().i2v_cross_attn_fused_supported(rows, channels, heads, head_dim, ctx_len, rows_per_ctx))


def pack_ctx_fragments(k, vt, heads, ctx_len, out=None):
    """`ctx_frag` of i2v_cross_attn_fused_f16 from the projected context as the other kernels take it -- k [n_ctx * ctx_len, C],
    vt [n_ctx, C, >= ctx_len] (V^T) -- as MFMA operand fragments [n_ctx][heads][30][64][4], zero beyond the context's length
    and the head's width (i2v_pack_ctx_fragments_f16; once per prompt with a `ProjectedContext`, else once per forward).
    `out`: a previous result to overwrite in place (a captured hipGraph keeps reading that memory for the next prompt)."""
    lib = _lib.load()
    k, ldk = _mat(k, "k")
    _req(vt, "vt")
    if vt.dim() != 3 or vt.stride(2) != 1:
        raise ValueError(f"pack_ctx_fragments: vt must be [n_ctx, C, >= ctx_len] with unit stride along the keys, got {tuple(vt.shape)}")
    n_ctx, c = vt.shape[0], vt.shape[1]
    d = c // heads
    dt, kt_n = (d + 15) // 16, 5
    if ctx_len > 16 * kt_n or k.shape[0] != n_ctx * ctx_len or k.shape[1] < c or vt.shape[2] < ctx_len or c % heads:
        raise ValueError(f"pack_ctx_fragments: k {tuple(k.shape)} / vt {tuple(vt.shape)} / ctx_len {ctx_len}")
    shape = (n_ctx, heads, 2 * kt_n * dt, 64, 4)
    if n_ctx * heads * 2 * kt_n * dt * 256 != lib.i2v_cross_attn_fused_ctx_elems(n_ctx, heads, d):
        raise RuntimeError("pack_ctx_fragments: size differs from i2v_cross_attn_fused_ctx_elems")
    if out is None:
        out = torch.empty(shape, dtype=f16, device=k.device)
    elif tuple(out.shape) != shape or out.dtype != f16 or not out.is_contiguous():
        raise ValueError(f"pack_ctx_fragments: out is {tuple(out.shape)}, expected {shape}")
    _lib.check(lib.i2v_pack_ctx_fragments_f16(_p(k), ldk, _p(vt), vt.stride(1), vt.stride(0), _p(out), n_ctx, heads, d, ctx_len,
                                              _stream()), "i2v_pack_ctx_fragments_f16")
    return out


def pack_attn_out(w_o, b_o, heads):
    """(w_o fragments, b_o fp32) of the out-projection inside i2v_motion_attn_f16 / i2v_cross_attn_fused_f16: to_out[0].weight's
    rows per head-sized slice in fragment order (the layout of `pack_cross_q`), the bias in fp32."""
    return pack_cross_q(w_o, heads), b_o.detach().float().contiguous()


def _attn_out_operands(out_proj, c, heads, head_dim, what):
    w_o, b_o = out_proj
    _req(w_o, "w_o")
    _req(b_o, "b_o", dtype=torch.float32)
    if tuple(w_o.shape) != (_lib.load().i2v_cross_attn_fused_pack_rows(heads, head_dim), c) or not w_o.is_contiguous() or \
            b_o.numel() != c or not b_o.is_contiguous():
        raise ValueError(f"{what}: out_proj is {tuple(w_o.shape)} / {tuple(b_o.shape)} (pack_attn_out)")
    return _p(w_o), _p(b_o)


def cross_attn_fused(x, gamma32, beta32, w_q, ctx_frag, *, heads, head_dim, ctx_len, rows_per_ctx, eps, scale=None, out=None,
                     ip_frag=None, ip_len=0, ip_scale=1.0, out_proj=None):
    """o = softmax((LayerNorm(x) Wq^T) K^T) V against a short context (i2v_cross_attn_fused_f16): x [rows, C]; ctx_frag from
    `pack_ctx_fragments`; rows [i * rows_per_ctx, (i + 1) * rows_per_ctx) use context i.  ip_frag (+ ip_len <= 16, ip_scale):
    the IP-Adapter's image tokens packed the same way; their softmax is added with weight ip_scale."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    rows, c = x.shape
    _req(w_q, "w_q")
    _req(ctx_frag, "ctx_frag")
    for name, t in (("gamma32", gamma32), ("beta32", beta32)):
        _req(t, name, dtype=torch.float32)
        if tuple(t.shape) != (c,):
            raise ValueError(f"cross_attn_fused: {name} is {tuple(t.shape)}, expected ({c},)")
    n_ctx = -(-rows // rows_per_ctx)
    if c != heads * head_dim or not ctx_frag.is_contiguous() or \
            ctx_frag.numel() != lib.i2v_cross_attn_fused_ctx_elems(n_ctx, heads, head_dim):
        raise ValueError(f"cross_attn_fused: ctx_frag is {tuple(ctx_frag.shape)} for {n_ctx} contexts (pack_ctx_fragments)")
    if tuple(w_q.shape) != (lib.i2v_cross_attn_fused_pack_rows(heads, head_dim), c) or not w_q.is_contiguous():
        raise ValueError(f"cross_attn_fused: w_q is {tuple(w_q.shape)} (pack_cross_q)")
    if out is None:
        out = torch.empty((rows, c), dtype=f16, device=x.device)
    out, ldo = _mat(out, "out")
    p = _lib.CrossAttnFusedParams()
    p.x, p.ldx = _p(x), ldx
    p.gamma, p.beta, p.w_q = _p(gamma32), _p(beta32), _p(w_q)
    p.ctx_frag = _p(ctx_frag)
    p.out, p.ldo = _p(out), ldo
    p.rows, p.rows_per_ctx = rows, rows_per_ctx
    p.channels, p.heads, p.head_dim, p.ctx_len = c, heads, head_dim, ctx_len
    p.eps, p.scale = float(eps), float(head_dim) ** -0.5 if scale is None else float(scale)
    if ip_frag is not None:
        _req(ip_frag, "ip_frag")
        if not ip_frag.is_contiguous() or ip_frag.numel() != ctx_frag.numel():
            raise ValueError(f"cross_attn_fused: ip_frag is {tuple(ip_frag.shape)} (pack_ctx_fragments of the image tokens)")
        p.ip_frag, p.ip_len, p.ip_scale = _p(ip_frag), int(ip_len), float(ip_scale)
    if out_proj is not None:      # out = x + o Wo^T + bo in the same launch (`pack_attn_out`)
        p.w_o, p.b_o = _attn_out_operands(out_proj, c, heads, head_dim, "cross_attn_fused")
    _lib.check(lib.i2v_cross_attn_fused_f16(C.byref(p), _stream()), "i2v_cross_attn_fused_f16")
    return out


def ln_qkv_supported(rows, channels, n_qk, rows_per_image):
    """is the one-launch LayerNorm + [q | k | q_adapter] + V^T projection implemented for this shape?"""
    return bool(_lib.load().i2v_ln_qkv_supported(rows, channels, n_qk, rows_per_image))


def pack_ln_qkv(w_qk, w_v):
    """`w` of i2v_ln_qkv_f16: the rows of [w_qk ; w_v] per 16-row tile in fragment order [tile][C / 32][lane][8]."""
    w = torch.cat([w_qk.detach().to(f16), w_v.detach().to(f16)], dim=0)
    n, c = w.shape
    if n % 160 or c % 32:
        raise ValueError(f"pack_ln_qkv: {n} rows x {c} columns is not whole waves of 160 rows / K steps of 32")
    return w.view(n // 16, 16, c // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(n, c)     # [tile, s, g, l15, j]


def ln_qkv(x, gamma32, beta32, w_packed, *, n_qk, rows_per_image, eps, qk=None, vt=None, images=None, x_image_stride=0):
    """(qk [rows, n_qk], vt [rows / rows_per_image, C, pad8(rows_per_image)]) = projections of LayerNorm(x) in one launch
    (i2v_ln_qkv_f16); w_packed = `pack_ln_qkv(w_qk, w_v)`.  images / x_image_stride: only `images` blocks of rows_per_image rows
    of x are taken, block i starting at row i * x_image_stride / ldx (the frame-0 rows of every clip, read in place)."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    rows, c = x.shape
    if images is not None:
        if x_image_stride % ldx or (images - 1) * (x_image_stride // ldx) + rows_per_image > rows:
            raise ValueError(f"ln_qkv: {images} images of {rows_per_image} rows at stride {x_image_stride} do not fit x {tuple(x.shape)}")
        rows = images * rows_per_image
    _req(w_packed, "w")
    for name, t in (("gamma32", gamma32), ("beta32", beta32)):
        _req(t, name, dtype=torch.float32)
    if tuple(gamma32.shape) != (c,) or tuple(beta32.shape) != (c,) or tuple(w_packed.shape) != (n_qk + c, c) or \
            not w_packed.is_contiguous():
        raise ValueError(f"ln_qkv: gamma {tuple(gamma32.shape)}, beta {tuple(beta32.shape)}, w {tuple(w_packed.shape)} for C {c}, "
                         f"n_qk {n_qk} (pack_ln_qkv)")
    if rows % rows_per_image:
        raise ValueError(f"ln_qkv: {rows} rows are not whole images of {rows_per_image}")
    n_img, ld = rows // rows_per_image, pad8(rows_per_image)
    if qk is None:
        qk = torch.empty((rows, n_qk), dtype=f16, device=x.device)
    if vt is None:
        vt = torch.empty((n_img, c, ld), dtype=f16, device=x.device)
    qk, ld_qk = _mat(qk, "qk")
    _req(vt, "vt")
    if tuple(qk.shape) != (rows, n_qk) or tuple(vt.shape) != (n_img, c, ld) or not vt.is_contiguous():
        raise ValueError(f"ln_qkv: qk {tuple(qk.shape)}, vt {tuple(vt.shape)}")
    p = _lib.LnQkvParams()
    p.x, p.ldx = _p(x), ldx
    p.gamma, p.beta, p.w = _p(gamma32), _p(beta32), _p(w_packed)
    p.qk, p.ld_qk = _p(qk), ld_qk
    p.vt, p.vt_batch_stride, p.vt_row_stride = _p(vt), c * ld, ld
    p.rows, p.rows_per_image, p.channels, p.n_qk, p.eps = rows, rows_per_image, c, n_qk, float(eps)
    p.x_image_stride = int(x_image_stride) if images is not None else 0
    _lib.check(lib.i2v_ln_qkv_f16(C.byref(p), _stream()), "i2v_ln_qkv_f16")
    return qk, vt


def ff_fused_supported(rows, channels, inner):
    """is the one-launch GEGLU feed-forward implemented for this shape?"""
    return bool(_lib.load().i2v_ff_fused_supported(rows, channels, inner))


def pack_ff_fused(w1, b1, w2, b2):
    """(w1 fragments, b1 fp32, w2 fragments, b2 fp32) of i2v_ff_fused_f16 from diffusers' GEGLU.proj (w1 [2 inner, C], rows
    [values ; gates]) and the output Linear (w2 [C, inner]); layouts in include/i2v_hip.h."""
    inner, c = w2.shape[1], w2.shape[0]
    nch, heads, dn = inner // 128, 8, c // 8
    dt = (dn + 15) // 16
    dev = w1.device
    # W1 rows of tile (ch, w, u): m = 0 .. 15 -> (m & 1) * inner + 128 ch + 16 w + 8 u + (m >> 1)
    m = torch.arange(16, device=dev)
    idx = ((m & 1) * inner + (m >> 1))[None, None, None, :] + (128 * torch.arange(nch, device=dev))[:, None, None, None] + \
        (16 * torch.arange(8, device=dev))[None, :, None, None] + (8 * torch.arange(2, device=dev))[None, None, :, None]   # [nch, 8, 2, 16]
    t1 = w1.detach().to(f16)[idx.reshape(-1)].view(nch, 8, 2, 16, c // 32, 4, 8)                      # [ch, w, u, l15, s, g, j]
    w1f = t1.permute(0, 1, 2, 4, 5, 3, 6).contiguous().view(-1, 8)                                     # [ch, w, u, s, g, l15, j]
    b1f = b1.detach().float()[idx.reshape(-1)].view(nch, 8, 2, 16).contiguous()
    w2p = torch.zeros((heads, 16 * dt, inner), dtype=f16, device=dev)
    w2p[:, :dn] = w2.detach().to(f16).view(heads, dn, inner)
    t2 = w2p.view(heads, dt, 16, nch, 4, 4, 8)                                                         # [w, t, l15, ch, ks, g, j]
    w2f = t2.permute(0, 3, 4, 1, 5, 2, 6).contiguous().view(-1, 8)                                     # [w, ch, ks, t, g, l15, j]
    return w1f, b1f, w2f, b2.detach().float().contiguous()


def ff_fused_tail_supported(rows, channels, inner, perm_frames=0, perm_hw=0):
    """... and with the Linear that follows the block in the same launch (`tail=` of ff_fused)?"""
    return bool(_lib.load().i2v_ff_fused_tail_supported(rows, channels, inner, perm_frames, perm_hw))


def pack_ff_tail(w3, b3):
    """(w3 fragments, b3 fp32) of the tail of i2v_ff_fused_f16: the [C, C] Linear per 40-row slice in fragment order."""
    return pack_cross_q(w3, 8), b3.detach().float().contiguous()


def ff_fused(x, gamma32, beta32, packed, *, eps, out=None, tail=None, precise=False):
    """out = x + GEGLU-FF(LayerNorm(x)) in one launch (i2v_ff_fused_f16); packed = `pack_ff_fused(...)`.
    tail = (packed_tail, res2, perm_frames, perm_hw): the block's proj_out in the same launch --
    out[perm(r)] = res2[perm(r)] + (x + FF(LN(x)))[r] W3^T + b3, packed_tail = `pack_ff_tail(w3, b3)`; perm_frames > 0: rows are
    in (batch, pixel, frame) order and leave in (batch, frame, pixel) order (res2 is read in that order too).
    precise (with a tail whose res2 carries a low half, `lo_of`): res2 + its low half is added and the result's low half is written."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    rows, c = x.shape
    w1f, b1f, w2f, b2f = packed
    inner = b1f.numel() // 2
    for name, t, dt_ in (("gamma32", gamma32, torch.float32), ("beta32", beta32, torch.float32), ("w1", w1f, f16),
                         ("b1", b1f, torch.float32), ("w2", w2f, f16), ("b2", b2f, torch.float32)):
        _req(t, name, dtype=dt_)
    if tuple(gamma32.shape) != (c,) or tuple(beta32.shape) != (c,) or b2f.numel() != c or w1f.numel() != 2 * inner * c or \
            w2f.numel() != 8 * ((c // 8 + 15) // 16 * 16) * inner:
        raise ValueError(f"ff_fused: operand sizes do not match C {c}, inner {inner} (pack_ff_fused)")
    if out is None:
        out = torch.empty((rows, c), dtype=f16, device=x.device)
    out, ldo = _mat(out, "out")
    if tuple(out.shape) != (rows, c):
        raise ValueError(f"ff_fused: out is {tuple(out.shape)}")
    p = _lib.FfFusedParams()
    p.x, p.ldx = _p(x), ldx
    p.gamma, p.beta = _p(gamma32), _p(beta32)
    p.w1, p.b1, p.w2, p.b2 = _p(w1f), _p(b1f), _p(w2f), _p(b2f)
    p.out, p.ldo = _p(out), ldo
    p.rows, p.channels, p.inner, p.eps = rows, c, inner, float(eps)
    keep = None
    if tail is not None:
        (w3f, b3f), res2, perm_frames, perm_hw = tail
        _req(w3f, "w3")
        _req(b3f, "b3", dtype=torch.float32)
        res2, ld2 = _mat(res2, "res2")
        if tuple(w3f.shape) != (8 * ((c // 8 + 15) // 16 * 16), c) or not w3f.is_contiguous() or b3f.numel() != c or \
                tuple(res2.shape) != (rows, c):
            raise ValueError(f"ff_fused: tail operands w3 {tuple(w3f.shape)}, b3 {tuple(b3f.shape)}, res2 {tuple(res2.shape)} "
                             f"do not match rows {rows}, C {c} (pack_ff_tail)")
        if perm_frames and out.data_ptr() == x.data_ptr():
            raise ValueError("ff_fused: out must not alias x when the tail permutes the rows")
        p.w3, p.b3, p.res2, p.ld_res2 = _p(w3f), _p(b3f), _p(res2), ld2
        p.perm_frames, p.perm_hw = int(perm_frames), int(perm_hw)
        if precise and lo_of(res2) is not None:
            def set_lo(rl, cl):
                p.res2_lo, p.out_lo = _p(rl), _p(cl)
            keep = _attach_lo(set_lo, res2, out)
    _lib.check(lib.i2v_ff_fused_f16(C.byref(p), _stream()), "i2v_ff_fused_f16")
    del keep
    return out


def motion_attn_tables(gamma, beta, pe, frames):
    """(gamma fp32 [C], shift fp32 [frames, C] = beta + pe[frame]): the LayerNorm constants of i2v_motion_attn_f16."""
    return (gamma.detach().float().contiguous(),
            (beta.detach().float()[None, :] + pe.detach().float()[:frames]).contiguous())


def motion_attn(x, gamma32, shift32, w_qkv, *, heads, head_dim, frames, eps, scale=None, out=None, out_proj=None):
    """o = temporal attention over the `frames` rows of each pixel of LayerNorm(x) + pe, q / k / v projected inside
    (i2v_motion_attn_f16); x [rows, C] in (batch, pixel, frame) order, (gamma32, shift32) from `motion_attn_tables`,
    w_qkv from `pack_motion_qkv`."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    rows, c = x.shape
    _req(w_qkv, "w_qkv")
    for name, t in (("gamma32", gamma32), ("shift32", shift32)):
        _req(t, name, dtype=torch.float32)
    if c != heads * head_dim or tuple(gamma32.shape) != (c,) or tuple(shift32.shape) != (frames, c) or not shift32.is_contiguous():
        raise ValueError(f"motion_attn: C {c} vs heads {heads} x head_dim {head_dim}, gamma {tuple(gamma32.shape)}, "
                         f"shift {tuple(shift32.shape)} (expected [{frames}, {c}])")
    if tuple(w_qkv.shape) != (lib.i2v_motion_attn_pack_rows(heads, head_dim), c) or not w_qkv.is_contiguous():
        raise ValueError(f"motion_attn: w_qkv is {tuple(w_qkv.shape)} (pack_motion_qkv)")
    if out is None:
        out = torch.empty((rows, c), dtype=f16, device=x.device)
    out, ldo = _mat(out, "out")
    if tuple(out.shape) != (rows, c):
        raise ValueError(f"motion_attn: out is {tuple(out.shape)}")
    p = _lib.MotionAttnParams()
    p.x, p.ldx = _p(x), ldx
    p.gamma = _p(gamma32)
    p.shift, p.ld_shift = _p(shift32), c
    p.w_qkv = _p(w_qkv)
    p.out, p.ldo = _p(out), ldo
    p.rows, p.channels, p.heads, p.head_dim, p.frames = rows, c, heads, head_dim, frames
    p.eps, p.scale = float(eps), float(head_dim) ** -0.5 if scale is None else float(scale)
    if out_proj is not None:      # out = x + o Wo^T + bo in the same launch (`pack_attn_out`)
        p.w_o, p.b_o = _attn_out_operands(out_proj, c, heads, head_dim, "motion_attn")
    _lib.check(lib.i2v_motion_attn_f16(C.byref(p), _stream()), "i2v_motion_attn_f16")
    return out


# ---------------------------------------------------------------------------------------------- norms
def groupnorm(x, gamma, beta, groups, eps, *, x2=None, silu=False, frames_per_stat=1, out_perm=False, frames=0, stats=None):
    """GroupNorm (+SiLU) of a token-major image batch x [N, H, W, C1] (optionally concatenated with x2 along C).
    Returns [N, H, W, C]; with out_perm the rows are written in (b, pixel, frame) order and the result is
    returned flat as [N * H * W, C]."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"x must be contiguous [N, H, W, C], got {tuple(x.shape)}")
    n, h, w, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _req(x2, "x2")
        if x2.dim() != 4 or not x2.is_contiguous() or x2.shape[:3] != x.shape[:3]:
            raise ValueError("x2 must be contiguous [N, H, W, C2] with the same N, H, W as x")
        c2 = x2.shape[3]
    Cc = c1 + c2
    _req(gamma, "gamma")
    _req(beta, "beta")
    if gamma.numel() != Cc or beta.numel() != Cc:
        raise ValueError(f"gamma / beta must have {Cc} elements")
    ws = torch.empty((lib.i2v_groupnorm_workspace_bytes(n, h * w, Cc) + 3) // 4, dtype=torch.float32, device=x.device)
    y = torch.empty((n * h * w, Cc) if out_perm else (n, h, w, Cc), dtype=f16, device=x.device)
    p = GnParams()
    p.x, p.c1, p.x2, p.c2 = _p(x), c1, _p(x2), c2
    p.gamma, p.beta, p.y = _p(gamma.contiguous()), _p(beta.contiguous()), _p(y)
    p.n_img, p.hw, p.groups, p.frames_per_stat = n, h * w, groups, frames_per_stat
    p.eps, p.silu = eps, 1 if silu else 0
    p.out_perm, p.frames = (1 if out_perm else 0), frames
    p.workspace = _p(ws)
    if stats is not None:       # (partials, rows per partial block) from `conv3x3(..., gn_stats_groups=groups)`: no statistics pass
        part, rows = stats
        _req(part, "stats", dtype=torch.float32)
        if x2 is not None or frames_per_stat != 1 or out_perm or tuple(part.shape) != (n, (h * w) // rows, groups, 2) or \
                not part.is_contiguous():
            raise ValueError(f"groupnorm: stats {tuple(part.shape)} / rows {rows} do not describe x {tuple(x.shape)} in {groups} groups")
        p.gpartial_in, p.gpartial_rows = _p(part), int(rows)
    _lib.check(lib.i2v_groupnorm_f16(C.byref(p), _stream()), "i2v_groupnorm_f16")
    return y


def groupnorm_fold(x, gamma, beta, groups, eps, w, bias, *, x2=None, frames_per_stat=1):
    """GroupNorm (no activation) of x [N, H, W, C1] (+ x2 along C) folded into the Linear (w [n_out, C], bias) that
    consumes it: returns (w_s [S, n_out, C], bias_s [S, n_out]) with S = N / frames_per_stat statistics groups, such that
    Linear(GroupNorm(x)) = x w_s[s]^T + bias_s[s] for the rows of group s (i2v_groupnorm_fold_f16)."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"x must be contiguous [N, H, W, C], got {tuple(x.shape)}")
    n, h, wd, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _req(x2, "x2")
        if x2.dim() != 4 or not x2.is_contiguous() or x2.shape[:3] != x.shape[:3]:
            raise ValueError("x2 must be contiguous [N, H, W, C2] with the same N, H, W as x")
        c2 = x2.shape[3]
    Cc = c1 + c2
    _req(gamma, "gamma")
    _req(beta, "beta")
    w2, ldw = _mat(w, "w")
    if gamma.numel() != Cc or beta.numel() != Cc or w2.shape[1] != Cc:
        raise ValueError(f"gamma / beta / w columns must have {Cc} elements")
    n_out = w2.shape[0]
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != n_out or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous vector of n_out elements")
    if frames_per_stat <= 0 or n % frames_per_stat != 0:
        raise ValueError(f"batch {n} is not a multiple of frames_per_stat {frames_per_stat}")
    S = n // frames_per_stat
    ws = torch.empty((lib.i2v_groupnorm_workspace_bytes(n, h * wd, Cc) + 3) // 4, dtype=torch.float32, device=x.device)
    w_s = torch.empty((S, n_out, Cc), dtype=f16, device=x.device)
    b_s = torch.empty((S, n_out), dtype=f16, device=x.device)
    p = GnParams()
    p.x, p.c1, p.x2, p.c2 = _p(x), c1, _p(x2), c2
    p.gamma, p.beta = _p(gamma.contiguous()), _p(beta.contiguous())
    p.n_img, p.hw, p.groups, p.frames_per_stat = n, h * wd, groups, frames_per_stat
    p.eps = eps
    p.workspace = _p(ws)
    _lib.check(lib.i2v_groupnorm_fold_f16(C.byref(p), _p(w2), ldw, _p(bias), n_out, _p(w_s), _p(b_s), _stream()),
               "i2v_groupnorm_fold_f16")
    return w_s, b_s


def layernorm(x, gamma, beta, eps, *, pe=None, pe_period=0):
    """LayerNorm over the last dim of a 2-D token matrix (+ pe[row % pe_period]).  x may also be a 3-D view
    [batches, rows_per_batch, C] with arbitrary batch stride (e.g. the frame-0 rows of every clip, i2v:484): the rows are
    read in place and the result is the dense [batches * rows_per_batch, C] matrix."""
    lib = _lib.load()
    batched = None
    if x.dim() == 3:
        _req(x, "x")
        if x.stride(2) != 1 or x.stride(1) % 8 != 0 or x.stride(0) % 8 != 0 or x.stride(0) <= 0:
            raise ValueError("batched layernorm input needs unit last stride and row / batch strides that are multiples of 8")
        batched = (x.shape[1], x.stride(0))
        ldx = x.stride(1)
        rows, Cc = x.shape[0] * x.shape[1], x.shape[2]
    else:
        x, ldx = _mat(x, "x")
        rows, Cc = x.shape
    _req(gamma, "gamma")
    _req(beta, "beta")
    y = torch.empty((rows, Cc), dtype=f16, device=x.device)
    p = LnParams()
    p.x, p.ldx = _p(x), ldx
    p.gamma, p.beta = _p(gamma), _p(beta)
    if pe is not None:
        pe, ldpe = _mat(pe, "pe")
        if pe.shape[1] != Cc or pe.shape[0] < pe_period or pe_period <= 0:
            raise ValueError("pe must be [>= pe_period, C]")
        p.pe, p.ld_pe, p.pe_period = _p(pe), ldpe, pe_period
    p.y, p.ldy = _p(y), Cc
    p.rows, p.C, p.eps = rows, Cc, eps
    if batched is not None:
        p.x_rows_per_batch, p.x_batch_stride = batched
    _lib.check(lib.i2v_layernorm_f16(C.byref(p), _stream()), "i2v_layernorm_f16")
    return y


def softmax_rows(x, scale=1.0, out=None):
    """row-wise softmax(scale * x) of a 2-D fp16 matrix (in place when out is x)."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    if out is None:                      # rows start on 16-byte boundaries: leading dimension rounded up to 8
        out = torch.empty((x.shape[0], pad8(x.shape[1])), dtype=f16, device=x.device)[:, : x.shape[1]]
    out, ldy = _mat(out, "out")
    if out.shape != x.shape:
        raise ValueError("softmax_rows: out must have x's shape")
    _lib.check(lib.i2v_softmax_rows_f16(_p(x), ldx, _p(out), ldy, x.shape[0], x.shape[1], float(scale), _stream()),
               "i2v_softmax_rows_f16")
    return out


# ---------------------------------------------------------------------------------------------- edges / misc
def nchw_to_tokens(src, c_pad=None):
    """[N, C, H, W] (fp32 or fp16) -> token-major fp16 [N, H, W, c_pad]."""
    lib = _lib.load()
    _req(src, "src", dtype=None)
    if src.dtype not in (torch.float32, f16) or src.dim() != 4:
        raise TypeError("src must be a 4-D fp32 / fp16 tensor")
    src = src.contiguous()
    n, c, h, w = src.shape
    c_pad = c if c_pad is None else c_pad
    dst = torch.empty((n, h, w, c_pad), dtype=f16, device=src.device)
    _lib.check(lib.i2v_nchw_to_tokens(_p(src), 1 if src.dtype == torch.float32 else 0, _p(dst), n, c, h * w, c_pad,
                                      _stream()), "i2v_nchw_to_tokens")
    return dst


def tokens_to_nchw(src, c=None, dtype=f16):
    """token-major fp16 / fp32 [N, H, W, ld] -> [N, c, H, W] in `dtype` (fp16 / fp32)."""
    lib = _lib.load()
    _req(src, "src", dtype=None)
    if src.dtype not in (torch.float32, f16):
        raise TypeError("src must be fp16 or fp32")
    if src.dim() != 4 or not src.is_contiguous():
        raise ValueError("src must be contiguous [N, H, W, C]")
    n, h, w, ld = src.shape
    c = ld if c is None else c
    if dtype not in (torch.float32, f16):
        raise TypeError("dtype must be fp32 or fp16")
    dst = torch.empty((n, c, h, w), dtype=dtype, device=src.device)
    _lib.check(lib.i2v_tokens_to_nchw(_p(src), 1 if src.dtype == torch.float32 else 0, ld, _p(dst),
                                      1 if dtype == torch.float32 else 0, n, c, h * w, _stream()), "i2v_tokens_to_nchw")
    return dst


def timestep_embedding(t, dim, t_index=None):
    """t fp32 [n] (or a table indexed by the device scalar `t_index`, with n = t_rows) -> [n, dim] fp16."""
    lib = _lib.load()
    _req(t, "t", dtype=torch.float32)
    n = 1 if t_index is not None else t.numel()
    if t_index is not None:
        _req(t_index, "t_index", dtype=torch.int32)
    out = torch.empty((n, dim), dtype=f16, device=t.device)
    _lib.check(lib.i2v_timestep_embedding(_p(t), _p(t_index), t.numel(), _p(out), n, dim, _stream()),
               "i2v_timestep_embedding")
    return out


def silu(x):
    lib = _lib.load()
    _req(x, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.check(lib.i2v_silu_f16(_p(x), _p(y), x.numel(), _stream()), "i2v_silu_f16")
    return y


def select_row(table, row_index, out=None):
    """[1, cols] = table[clamp(*row_index)] for a device int32 scalar `row_index` (a replayed step's row of a per-timestep
    table)."""
    lib = _lib.load()
    table, ld = _mat(table, "table")
    _req(row_index, "row_index", dtype=torch.int32)
    if out is None:
        out = torch.empty((1, table.shape[1]), dtype=f16, device=table.device)
    _lib.check(lib.i2v_select_row_f16(_p(table), ld, table.shape[0], _p(row_index), _p(out), table.shape[1], _stream()),
               "i2v_select_row_f16")
    return out


def repeat_rows(x, repeat):
    lib = _lib.load()
    x, _ = _mat(x, "x")
    x = x.contiguous()
    y = torch.empty((x.shape[0] * repeat, x.shape[1]), dtype=f16, device=x.device)
    _lib.check(lib.i2v_repeat_rows_f16(_p(x), _p(y), x.shape[0], x.shape[1], repeat, _stream()), "i2v_repeat_rows_f16")
    return y


def copy3d(src, dst):
    """dst[b, r, :] = src[b, r, :] for 3-D fp16 tensors whose last dim is unit-stride (views allowed)."""
    lib = _lib.load()
    _req(src, "src")
    _req(dst, "dst")
    if src.dim() != 3 or dst.shape != src.shape or src.stride(2) != 1 or dst.stride(2) != 1:
        raise ValueError("copy3d needs two 3-D tensors of equal shape with unit last stride")
    b, r, c = src.shape
    _lib.check(lib.i2v_copy3d_f16(_p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), b, r, c,
                                  _stream()), "i2v_copy3d_f16")
    return dst


def duplicate_batch(x):
    """[x ; x] along the leading dimension (the second CFG half of a tensor computed once for both, unet._fwd_tokens)."""
    _req(x, "x")
    lo = lo_of(x)
    x = x.contiguous()
    rows = x.shape[0]
    y = torch.empty((2 * rows,) + tuple(x.shape[1:]), dtype=f16, device=x.device)
    c = x.shape[-1]
    copy3d(x.view(1, -1, c).expand(2, -1, c), y.view(2, -1, c))
    if lo is not None:                      # a tensor of the precise residual stream: its low half travels with it
        y._i2v_lo = duplicate_batch(lo)
    return y


def first_frame_prior(cond, mask_uniform, noise, sigma, strength, sqrt_alpha, sqrt_one_minus_alpha):
    """latents = sqrt_alpha * (mask * blur3x3_sigma(cond) + (1 - mask) * cond) + sqrt_one_minus_alpha * noise with
    mask = (mask_uniform < strength): the first-frame-similarity prior + add_noise of pipe:647-656 in one kernel.
    cond fp32 [B, C, H, W]; mask_uniform / noise fp32 [B, F, C, H, W]; returns fp32 [B, F, C, H, W]."""
    import math
    lib = _lib.load()
    for t, name in ((cond, "cond"), (mask_uniform, "mask_uniform"), (noise, "noise")):
        _req(t, name, dtype=torch.float32)
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    if cond.dim() != 4 or noise.dim() != 5 or mask_uniform.shape != noise.shape or \
            (noise.shape[0], noise.shape[2], noise.shape[3], noise.shape[4]) != tuple(cond.shape):
        raise ValueError(f"cond {tuple(cond.shape)} / mask {tuple(mask_uniform.shape)} / noise {tuple(noise.shape)} mismatch")
    if not sigma > 0:
        raise ValueError("sigma must be positive")
    b, f, c, h, w = noise.shape
    e = math.exp(-0.5 / (sigma * sigma))                      # torchvision _get_gaussian_kernel1d, kernel_size 3
    kc, ke = 1.0 / (1.0 + 2.0 * e), e / (1.0 + 2.0 * e)
    out = torch.empty_like(noise)
    _lib.check(lib.i2v_first_frame_prior_f32(_p(cond), _p(mask_uniform), _p(noise), _p(out), b, f, c, h, w, kc, ke,
                                             float(strength), float(sqrt_alpha), float(sqrt_one_minus_alpha),
                                             _stream()), "i2v_first_frame_prior_f32")
    return out


def gaussian_sample(moments, eps):
    """mean + exp(0.5 clamp(logvar, -30, 20)) * eps for moments fp32 [N, 2C, H, W] = (mean | logvar), eps fp32 [N, C, H, W]."""
    lib = _lib.load()
    _req(moments, "moments", dtype=torch.float32)
    _req(eps, "eps", dtype=torch.float32)
    n, c2, h, w = moments.shape
    if c2 % 2 != 0 or tuple(eps.shape) != (n, c2 // 2, h, w) or not moments.is_contiguous() or not eps.is_contiguous():
        raise ValueError(f"moments {tuple(moments.shape)} / eps {tuple(eps.shape)} mismatch")
    out = torch.empty_like(eps)
    _lib.check(lib.i2v_gaussian_sample_f32(_p(moments), _p(eps), _p(out), n, c2 // 2, h * w, _stream()),
               "i2v_gaussian_sample_f32")
    return out


def ddim_prep(latents, cond, c_pad, cfg_copies):
    """latents fp32 [B, F, C, H, W] (frame 0 overwritten in place with cond [B, C, H, W]) ->
    model input tokens fp16 [cfg_copies * B * F, H, W, c_pad]."""
    lib = _lib.load()
    _req(latents, "latents", dtype=torch.float32)
    _req(cond, "cond", dtype=torch.float32)
    if latents.dim() != 5 or not latents.is_contiguous() or not cond.is_contiguous():
        raise ValueError("latents must be contiguous [B, F, C, H, W]; cond contiguous [B, C, H, W]")
    b, f, c, h, w = latents.shape
    if tuple(cond.shape) != (b, c, h, w):
        raise ValueError(f"cond must be {(b, c, h, w)}, got {tuple(cond.shape)}")
    out = torch.empty((cfg_copies * b * f, h, w, c_pad), dtype=f16, device=latents.device)
    _lib.check(lib.i2v_ddim_prep(_p(latents), _p(cond), _p(out), b, f, c, h * w, c_pad, cfg_copies, _stream()),
               "i2v_ddim_prep")
    return out


def ddim_cfg_step(latents, noise_pred, coef, step_index, guidance_scale, cfg_copies):
    """In-place DDIM update of latents fp32 [B, F, C, H, W] from noise_pred tokens (fp32, or fp16)
    [cfg_copies * B * F, H, W, ld]; coef fp32 [steps, 4]; step_index device int32 scalar (advanced by one, wrapping
    to 0 at the end of the table)."""
    lib = _lib.load()
    _req(latents, "latents", dtype=torch.float32)
    _req(noise_pred, "noise_pred", dtype=None)
    if noise_pred.dtype not in (torch.float32, f16):
        raise TypeError("noise_pred must be fp32 or fp16")
    _req(coef, "coef", dtype=torch.float32)
    _req(step_index, "step_index", dtype=torch.int32)
    b, f, c, h, w = latents.shape
    if noise_pred.dim() != 4 or not noise_pred.is_contiguous() or noise_pred.shape[0] != cfg_copies * b * f or \
            noise_pred.shape[1] != h or noise_pred.shape[2] != w or noise_pred.shape[3] < c:
        raise ValueError(f"noise_pred must be contiguous [{cfg_copies * b * f}, {h}, {w}, >={c}]")
    if coef.dim() != 2 or coef.shape[1] != 4 or not coef.is_contiguous():
        raise ValueError("coef must be contiguous [steps, 4]")
    _lib.check(lib.i2v_ddim_cfg_step(_p(latents), _p(noise_pred), 1 if noise_pred.dtype == torch.float32 else 0,
                                     noise_pred.shape[3], _p(coef), coef.shape[0],
                                     _p(step_index), float(guidance_scale), b, f, c, h * w, cfg_copies, _stream()),
               "i2v_ddim_cfg_step")
    return latents


# ---------------------------------------------------------------------------------------------- backward
def transpose_tokens(x, batch_len, out=None):
    """[B * L, C] token-major -> [B, C, pad8(L)] channel-major (zero-filled pad): the K^T / Q^T / dO^T operands of the
    attention backward and the operands of a weight gradient (i2v_transpose_f16).  x may be a column slice."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    rows, c = x.shape
    if rows % batch_len != 0:
        raise ValueError(f"{rows} rows do not split into batches of {batch_len}")
    b, lp = rows // batch_len, pad8(batch_len)
    if out is None:
        out = torch.empty((b, c, lp), dtype=f16, device=x.device)
    _req(out, "out")
    if tuple(out.shape) != (b, c, lp) or not out.is_contiguous():
        raise ValueError(f"out must be contiguous [{b}, {c}, {lp}]")
    _lib.check(lib.i2v_transpose_f16(_p(x), batch_len * ldx, ldx, _p(out), c * lp, lp, b, batch_len, c, _stream()),
               "i2v_transpose_f16")
    return out


def attention_lse(q, k, *, batch_q, lq, lk, heads, head_dim, kv_group=1, scale=None):
    """fp32 [batch_q, heads, lq] log2-sum-exp of the forward attention over (q, k) (i2v_attention_lse_f32)."""
    lib = _lib.load()
    q, ldq = _mat(q, "q")
    k, ldk = _mat(k, "k")
    Cc, bkv = heads * head_dim, batch_q // kv_group
    if q.shape[0] != batch_q * lq or q.shape[1] < Cc or k.shape[0] != bkv * lk or k.shape[1] < Cc:
        raise ValueError(f"q {tuple(q.shape)} / k {tuple(k.shape)} do not match batch_q {batch_q}, lq {lq}, lk {lk}")
    lse = torch.empty((batch_q, heads, lq), dtype=torch.float32, device=q.device)
    p = AttnParams()
    p.q, p.q_row_stride, p.q_batch_stride = _p(q), ldq, lq * ldq
    p.k, p.k_row_stride, p.k_batch_stride = _p(k), ldk, lk * ldk
    p.batch_q, p.kv_group, p.heads, p.head_dim, p.lq, p.lk = batch_q, kv_group, heads, head_dim, lq, lk
    p.scale = float(head_dim) ** -0.5 if scale is None else scale
    _lib.check(lib.i2v_attention_lse_f32(C.byref(p), _p(lse), _stream()), "i2v_attention_lse_f32")
    return lse


def rowdot_heads(a, b, *, rows_per_batch, heads, head_dim):
    """fp32 [batches, heads, rows_per_batch]: sum over each head's channels of a * b (delta = rowsum(dO o O))."""
    lib = _lib.load()
    a, lda = _mat(a, "a")
    b, ldb = _mat(b, "b")
    rows = a.shape[0]
    if b.shape[0] != rows or rows % rows_per_batch != 0 or min(a.shape[1], b.shape[1]) < heads * head_dim:
        raise ValueError("rowdot_heads: shape mismatch")
    out = torch.empty((rows // rows_per_batch, heads, rows_per_batch), dtype=torch.float32, device=a.device)
    _lib.check(lib.i2v_rowdot_heads_f32(_p(a), lda, _p(b), ldb, _p(out), rows, rows_per_batch, heads, head_dim, _stream()),
               "i2v_rowdot_heads_f32")
    return out


def attention_bwd(q, k, v, o, dout, *, batch_q, lq, lk, heads, head_dim, kv_group=1, scale=None, need_dkv=True, lse=None):
    """Gradients of softmax(scale q k^T) v with respect to q (and k, v when need_dkv): token-major fp16 matrices in, the
    channel-major operand copies, the log-sum-exp (unless `lse` from the forward pass is given) and delta are made here.  Returns (dq, dk, dv); dk / dv [batch_kv * lk, C]
    are summed over the kv_group batch entries that share k / v (the frames of a clip for the cross-frame attention)."""
    lib = _lib.load()
    q, ldq = _mat(q, "q")
    k, ldk = _mat(k, "k")
    v, ldv = _mat(v, "v")
    o, _ = _mat(o, "o")
    dout, ldo = _mat(dout, "dout")
    Cc, bkv = heads * head_dim, batch_q // kv_group
    sc = float(head_dim) ** -0.5 if scale is None else scale
    if lse is None:   # (the forward pass can hand it over: attention(..., return_lse=True))
        lse = attention_lse(q, k, batch_q=batch_q, lq=lq, lk=lk, heads=heads, head_dim=head_dim, kv_group=kv_group, scale=sc)
    elif lse.dtype != torch.float32 or tuple(lse.shape) != (batch_q, heads, lq) or not lse.is_contiguous():
        raise ValueError(f"lse must be contiguous fp32 [{batch_q}, {heads}, {lq}]")
    delta = rowdot_heads(dout, o, rows_per_batch=lq, heads=heads, head_dim=head_dim)
    kt = transpose_tokens(k[:, :Cc], lk)
    dq = torch.empty((batch_q * lq, Cc), dtype=f16, device=q.device)
    p = AttnBwdParams()
    p.q, p.q_row_stride, p.q_batch_stride = _p(q), ldq, lq * ldq
    p.k, p.k_row_stride, p.k_batch_stride = _p(k), ldk, lk * ldk
    p.v, p.v_row_stride, p.v_batch_stride = _p(v), ldv, lk * ldv
    p.kt, p.kt_row_stride, p.kt_batch_stride = _p(kt), kt.shape[2], Cc * kt.shape[2]
    p.dout, p.do_row_stride, p.do_batch_stride = _p(dout), ldo, lq * ldo
    p.lse, p.delta = _p(lse), _p(delta)
    p.dq, p.dq_row_stride, p.dq_batch_stride = _p(dq), Cc, lq * Cc
    dk = dv = qt = dot = None
    if need_dkv:
        qt, dot = transpose_tokens(q[:, :Cc], lq), transpose_tokens(dout[:, :Cc], lq)
        dk = torch.empty((bkv * lk, Cc), dtype=f16, device=q.device)
        dv = torch.empty((bkv * lk, Cc), dtype=f16, device=q.device)
        p.qt, p.qt_row_stride, p.qt_batch_stride = _p(qt), qt.shape[2], Cc * qt.shape[2]
        p.doutt, p.dot_row_stride, p.dot_batch_stride = _p(dot), dot.shape[2], Cc * dot.shape[2]
        p.dk, p.dk_row_stride, p.dk_batch_stride = _p(dk), Cc, lk * Cc
        p.dv, p.dv_row_stride, p.dv_batch_stride = _p(dv), Cc, lk * Cc
    p.batch_q, p.kv_group, p.heads, p.head_dim, p.lq, p.lk = batch_q, kv_group, heads, head_dim, lq, lk
    p.scale = sc
    parts = dkv_partitions(batch_q, kv_group, heads, head_dim, lq, lk) if need_dkv else 1
    if parts > 1:   # the cross-frame form: too few (key block, head) workgroups for the chip; the frames are dealt out
        ws = torch.empty((2, parts, bkv * lk, Cc), dtype=torch.float32, device=q.device)
        p.kv_partitions, p.dkv_partial = parts, _p(ws)
    _lib.check(lib.i2v_attention_bwd_f16(C.byref(p), _stream()), "i2v_attention_bwd_f16")
    return dq, dk, dv


def dkv_partitions(batch_q, kv_group, heads, head_dim, lq, lk):
    """how many workgroups share the frames of one K / V in the dK / dV sweep (1 = off): a power of two dividing kv_group that
    brings the sweep to about one thousand workgroups."""
    if os.environ.get("I2V_ATTN_BWD_PARTS", "1") == "0" or kv_group < 2 or lq < 32:
        return 1
    two_k = lk >= 512 and head_dim <= 96                      # (the sweep's own choice of 128 or 64 keys per workgroup)
    blocks = ((lk + 127) // 128 if two_k else (lk + 63) // 64) * heads * (batch_q // kv_group)
    parts = 1
    while parts * 2 <= min(kv_group, 8) and kv_group % (parts * 2) == 0 and blocks * parts < 1024:
        parts *= 2
    return parts


def layernorm_bwd(x, dn, gamma, eps, add=None):
    """input gradient of LayerNorm (frozen affine): dx = LN'(x)[dn o gamma] (+ add, the residual path's gradient)."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    dn, lddn = _mat(dn, "dn")
    _req(gamma, "gamma")
    rows, c = x.shape
    if tuple(dn.shape) != (rows, c) or gamma.numel() != c:
        raise ValueError("layernorm_bwd: shape mismatch")
    lda = 0
    if add is not None:
        add, lda = _mat(add, "add")
        if tuple(add.shape) != (rows, c):
            raise ValueError("layernorm_bwd: add shape mismatch")
    dx = torch.empty((rows, c), dtype=f16, device=x.device)
    _lib.check(lib.i2v_layernorm_bwd_f16(_p(x), ldx, _p(dn), lddn, _p(gamma.contiguous()), _p(add), lda, _p(dx), c, rows, c,
                                         float(eps), _stream()), "i2v_layernorm_bwd_f16")
    return dx


def geglu(h):
    """y [rows, inner] = value * gelu_erf(gate) from the interleaved (value, gate) pre-activation h [rows, 2 inner]
    (i2v_geglu_f16: what the I2V_EPI_GEGLU epilogue applies, run on a stored h)."""
    lib = _lib.load()
    h, ldh = _mat(h, "h")
    rows, two_inner = h.shape
    if two_inner % 16 != 0:
        raise ValueError("geglu: the inner width must be a multiple of 8")
    y = torch.empty((rows, two_inner // 2), dtype=f16, device=h.device)
    _lib.check(lib.i2v_geglu_f16(_p(h), ldh, _p(y), two_inner // 2, rows, two_inner // 2, _stream()), "i2v_geglu_f16")
    return y


def geglu_bwd(h, dy):
    """h [rows, 2 inner] interleaved (value, gate) pre-activation, dy [rows, inner] -> dh [rows, 2 inner] (same order)."""
    lib = _lib.load()
    h, ldh = _mat(h, "h")
    dy, lddy = _mat(dy, "dy")
    rows, two_inner = h.shape
    if dy.shape[0] != rows or dy.shape[1] * 2 != two_inner:
        raise ValueError("geglu_bwd: shape mismatch")
    dh = torch.empty((rows, two_inner), dtype=f16, device=h.device)
    _lib.check(lib.i2v_geglu_bwd_f16(_p(h), ldh, _p(dy), lddy, _p(dh), two_inner, rows, two_inner // 2, _stream()),
               "i2v_geglu_bwd_f16")
    return dh


_colsum_ws = {}


def _colsum_workspace(lib, rows, cols, device):
    """scratch of the fixed-order column sums, one per device, grown on demand (stream-ordered use)"""
    need = lib.i2v_colsum_workspace_bytes(rows, cols)
    ws = _colsum_ws.get(device)
    if ws is None or ws.numel() * 4 < need:
        ws = _colsum_ws[device] = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=device)
    return ws


def colsum(x, out=None):
    """fp32 [cols] += sum over the rows of fp16 x (bias gradient); block sums added in a fixed order: bit-reproducible."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    if out is None:
        out = torch.zeros((x.shape[1],), dtype=torch.float32, device=x.device)
    _req(out, "out", dtype=torch.float32)
    ws = _colsum_workspace(lib, x.shape[0], x.shape[1], x.device)
    _lib.check(lib.i2v_colsum_det_f32(_p(x), ldx, None, 0, _p(out), x.shape[0], x.shape[1], _p(ws), _stream()), "i2v_colsum_det_f32")
    return out


def colsum_prod(a, b, out=None):
    """fp32 [cols] += sum over the rows of a o b (fp16 matrices of one shape): the gain gradient of a norm; fixed order."""
    lib = _lib.load()
    a, lda = _mat(a, "a")
    b, ldb = _mat(b, "b")
    if a.shape != b.shape:
        raise ValueError(f"colsum_prod: {tuple(a.shape)} vs {tuple(b.shape)}")
    if out is None:
        out = torch.zeros((a.shape[1],), dtype=torch.float32, device=a.device)
    _req(out, "out", dtype=torch.float32)
    ws = _colsum_workspace(lib, a.shape[0], a.shape[1], a.device)
    _lib.check(lib.i2v_colsum_det_f32(_p(a), lda, _p(b), ldb, _p(out), a.shape[0], a.shape[1], _p(ws), _stream()), "i2v_colsum_det_f32")
    return out


def masked_mse_grad(y, target, frames, coef):
    """seed gradient of the training loss (train_image_to_video.py:848-856): coef * (y - target) on the tokens of every
    frame but the first of each clip, 0 there.  y, target fp16 [n_img, tokens, C] contiguous."""
    lib = _lib.load()
    _req(y, "y")
    _req(target, "target")
    if y.dim() != 3 or y.shape != target.shape or not y.is_contiguous() or not target.is_contiguous():
        raise ValueError("masked_mse_grad: y / target must be equal-shape contiguous [n_img, tokens, C]")
    g = torch.empty_like(y)
    _lib.check(lib.i2v_masked_mse_grad_f16(_p(y), _p(target), _p(g), y.shape[0], y.shape[1], y.shape[2], frames, float(coef),
                                           _stream()), "i2v_masked_mse_grad_f16")
    return g


def masked_mse_grad_f32(y, target, frames, coef):
    """the same seed from an fp32 prediction and target [n_img, tokens, C] (train_image_to_video.py:848: the loss is taken in
    fp32).  Returns (fp16 gradient, fp32 [n_img * tokens] row sums of (y - target)^2 over the unmasked rows)."""
    lib = _lib.load()
    _req(y, "y", dtype=torch.float32)
    _req(target, "target", dtype=torch.float32)
    if y.dim() != 3 or y.shape != target.shape or not y.is_contiguous() or not target.is_contiguous():
        raise ValueError("masked_mse_grad_f32: y / target must be equal-shape contiguous [n_img, tokens, C]")
    g = torch.empty(y.shape, dtype=f16, device=y.device)
    rowsq = torch.empty((y.shape[0] * y.shape[1],), dtype=torch.float32, device=y.device)
    _lib.check(lib.i2v_masked_mse_grad_f32(_p(y), _p(target), _p(g), _p(rowsq), y.shape[0], y.shape[1], y.shape[2], frames,
                                           float(coef), _stream()), "i2v_masked_mse_grad_f32")
    return g, rowsq


def groupnorm_bwd(x, dy, gamma, beta, groups, eps, *, x2=None, silu=False, frames_per_stat=1):
    """input gradient of groupnorm(x [, x2], ...): dy [N, H, W, C1 + C2] -> dx [N, H, W, C1] (, dx2 [N, H, W, C2])."""
    lib = _lib.load()
    _req(x, "x")
    _req(dy, "dy")
    if x.dim() != 4 or not x.is_contiguous() or not dy.is_contiguous():
        raise ValueError("x / dy must be contiguous [N, H, W, C]")
    n, h, w, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _req(x2, "x2")
        if x2.dim() != 4 or not x2.is_contiguous() or x2.shape[:3] != x.shape[:3]:
            raise ValueError("x2 must be contiguous [N, H, W, C2] with the same N, H, W as x")
        c2 = x2.shape[3]
    Cc = c1 + c2
    if tuple(dy.shape) != (n, h, w, Cc) or gamma.numel() != Cc or beta.numel() != Cc:
        raise ValueError("groupnorm_bwd: shape mismatch")
    ws = torch.empty((lib.i2v_groupnorm_bwd_workspace_bytes(n, h * w, Cc) + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    p = GnParams()
    p.x, p.c1, p.x2, p.c2 = _p(x), c1, _p(x2), c2
    p.gamma, p.beta = _p(_req(gamma, "gamma").contiguous()), _p(_req(beta, "beta").contiguous())
    p.n_img, p.hw, p.groups, p.frames_per_stat = n, h * w, groups, frames_per_stat
    p.eps, p.silu = eps, 1 if silu else 0
    p.workspace = _p(ws)
    _lib.check(lib.i2v_groupnorm_bwd_f16(C.byref(p), _p(dy), _p(dx), _p(dx2), _stream()), "i2v_groupnorm_bwd_f16")
    return (dx, dx2) if x2 is not None else dx


def add(a, b, out=None):
    """a + b (fp16, same shape, contiguous)."""
    lib = _lib.load()
    _req(a, "a")
    _req(b, "b")
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous() or a.numel() % 8 != 0:
        raise ValueError("add: equal-shape contiguous tensors with a multiple of 8 elements expected")
    if out is None:
        out = torch.empty_like(a)
    _lib.check(lib.i2v_add_f16(_p(a), _p(b), _p(out), a.numel(), _stream()), "i2v_add_f16")
    return out


def permute_rows(x, batches, frames, hw, to_pixel_major):
    """[batches * frames * hw, C] rows (b, f, p) -> (b, p, f) (to_pixel_major) or back."""
    lib = _lib.load()
    x, ldx = _mat(x, "x")
    if x.shape[0] != batches * frames * hw or ldx != x.shape[1]:
        raise ValueError("permute_rows: contiguous [batches * frames * hw, C] expected")
    y = torch.empty_like(x)
    _lib.check(lib.i2v_permute_rows_f16(_p(x), _p(y), batches, frames, hw, x.shape[1], 1 if to_pixel_major else 0, _stream()),
               "i2v_permute_rows_f16")
    return y


def zero_insert2x(x):
    """[N, H, W, C] -> [N, 2H, 2W, C] with x at the even positions and zeros elsewhere."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("zero_insert2x: contiguous [N, H, W, C] expected")
    n, h, w, c = x.shape
    y = torch.empty((n, 2 * h, 2 * w, c), dtype=f16, device=x.device)
    _lib.check(lib.i2v_zero_insert2x_f16(_p(x), _p(y), n, h, w, c, _stream()), "i2v_zero_insert2x_f16")
    return y


def sum_pool2x(x):
    """[N, 2H, 2W, C] -> [N, H, W, C]: sums of the 2 x 2 blocks."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous() or x.shape[1] % 2 or x.shape[2] % 2:
        raise ValueError("sum_pool2x: contiguous [N, 2H, 2W, C] expected")
    n, h2, w2, c = x.shape
    y = torch.empty((n, h2 // 2, w2 // 2, c), dtype=f16, device=x.device)
    _lib.check(lib.i2v_sum_pool2x_f16(_p(x), _p(y), n, h2 // 2, w2 // 2, c, _stream()), "i2v_sum_pool2x_f16")
    return y


def sumsq(x, out=None):
    """fp32 [1] += sum of squares of a flat fp32 tensor (the global gradient norm)."""
    lib = _lib.load()
    _req(x, "x", dtype=torch.float32)
    if not x.is_contiguous():
        raise ValueError("sumsq: contiguous tensor expected")
    if out is None:
        out = torch.zeros((1,), dtype=torch.float32, device=x.device)
    _lib.check(lib.i2v_sumsq_f32(_p(x), x.numel(), _p(out), _stream()), "i2v_sumsq_f32")
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, *, lr, betas, eps, weight_decay, step, grad_coef=1.0, norm_sq=None,
               max_norm=0.0):
    """in-place AdamW update of a flat fp32 bucket (i2v_adamw_f32)."""
    lib = _lib.load()
    for t, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, name, dtype=torch.float32)
        if not t.is_contiguous() or t.numel() != param.numel():
            raise ValueError(f"adamw_step: {name} must be contiguous with {param.numel()} elements")
    _lib.check(lib.i2v_adamw_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
                                 float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_coef), _p(norm_sq),
                                 float(max_norm), _stream()), "i2v_adamw_f32")
    return param


def adamw_guarded_step(param, grad, exp_avg, exp_avg_sq, *, lr, betas, eps, weight_decay, grad_coef, max_norm, partials,
                       norm_sq, applied_steps, found_inf):
    """clip_grad_norm_ + AdamW of a flat fp32 bucket in one call, deterministic and overflow-safe (i2v_adamw_guarded_f32):
    `partials` fp32 [<= 1024] scratch, `norm_sq` fp32 [1] (out: sum grad^2), `applied_steps` int32 [1] (device step counter,
    advanced only by an applied step), `found_inf` int32 [1] (out: 1 = the gradients held inf / NaN and NOTHING was updated)."""
    lib = _lib.load()
    for t, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, name, dtype=torch.float32)
        if not t.is_contiguous() or t.numel() != param.numel():
            raise ValueError(f"adamw_guarded_step: {name} must be contiguous with {param.numel()} elements")
    _req(partials, "partials", dtype=torch.float32)
    _req(norm_sq, "norm_sq", dtype=torch.float32)
    _req(applied_steps, "applied_steps", dtype=torch.int32)
    _req(found_inf, "found_inf", dtype=torch.int32)
    if not partials.is_contiguous() or not 1 <= partials.numel() <= 1024:
        raise ValueError("adamw_guarded_step: partials must be a contiguous fp32 vector of 1 .. 1024 elements")
    _lib.check(lib.i2v_adamw_guarded_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr),
                                         float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(grad_coef),
                                         float(max_norm), _p(partials), partials.numel(), _p(norm_sq), _p(applied_steps),
                                         _p(found_inf), _stream()), "i2v_adamw_guarded_f32")
    return param


def axpby(y, x, a, b):
    """y = a y + b x in place over flat fp32 tensors (i2v_axpby_f32: gradient accumulation, EMA of the trained weights)."""
    lib = _lib.load()
    _req(y, "y", dtype=torch.float32)
    _req(x, "x", dtype=torch.float32)
    if not y.is_contiguous() or not x.is_contiguous() or y.numel() != x.numel():
        raise ValueError("axpby: two contiguous fp32 tensors of equal size expected")
    _lib.check(lib.i2v_axpby_f32(_p(y), _p(x), float(a), float(b), y.numel(), _stream()), "i2v_axpby_f32")
    return y

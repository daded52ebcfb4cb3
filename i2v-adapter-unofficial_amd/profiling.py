"""Live per-kernel-class timing with HIP events on the launch stream (bench.py's `roofline` object).

`with KernelProfile() as prof:` wraps every C-ABI wrapper of `kernels` with a pair of events recorded on torch's
current stream (the stream the ctypes launches go to) and the ALGORITHMIC work of the call:
  flops: gemm 2 M N K; conv3x3 2 M Cout 9 Cin; attention 4 Lq Lk C Bq; temporal attention 4 F F C pixels; fused motion
         attention sub-block 2 M C 3C + 4 M F C; fused text cross-attention 2 M C C + 4 M Lk C
  bytes: operands read once + result written once (fp16), GroupNorm 2 reads + 1 write, LayerNorm 1 read + 1 write
(the counting rules of SURVEY.md Appendix B / section 8d).  Only used outside the timed region.
"""
import torch

from . import kernels as K


def _numel_bytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if isinstance(t, torch.Tensor))


def _work_gemm(args, kw, out):
    a, w = args[0], args[1]
    M, N, Kd = a.shape[0], w.shape[-2], w.shape[-1]      # w may be a stack [S, N, K] of per-image weights (GroupNorm fold)
    tag = "".join(t for t, on in ((" geglu", kw.get("epilogue") == K.I2V_EPI_GEGLU), (" gelu", kw.get("epilogue") == K.I2V_EPI_GELU),
                                  (" +res", kw.get("residual") is not None), (" +ln", kw.get("ln") is not None),
                                  (" st%d" % kw.get("store", 0), bool(kw.get("store")))) if on)
    if w.dim() == 3:
        tag += " wstack%d" % w.shape[0] + (" perm" if kw.get("a_perm") is not None else "")
    return ("gemm", 2.0 * M * N * Kd, _numel_bytes(a, kw.get("a2"), w, kw.get("residual"), out),
            f"{M}x{N}x{Kd}{tag}")


def _work_conv(args, kw, out):
    x, w = args[0], args[1]
    stats = None
    if isinstance(out, tuple):        # (result, GroupNorm partials from the epilogue or None): gn_stats_groups
        out, stats = out
    M = out.shape[0] * out.shape[1] * out.shape[2]
    return ("conv3x3", 2.0 * M * w.shape[0] * w.shape[1], _numel_bytes(x, w, kw.get("residual"), out),
            f"{M}x{w.shape[0]}x{w.shape[1]}" + (" s2" if kw.get("stride", 1) == 2 else "")
            + (" up" if kw.get("upsample") else "") + (" +res" if kw.get("residual") is not None else "")
            + (" +gn stats" if stats is not None else ""))


def _work_attn(args, kw, out):
    c = kw["heads"] * kw["head_dim"]
    return ("attention", 4.0 * kw["batch_q"] * kw["lq"] * kw["lk"] * c, _numel_bytes(args[0], args[1], args[2], out),
            f"B{kw['batch_q']} Lq{kw['lq']} Lk{kw['lk']} h{kw['heads']} d{kw['head_dim']}")


def _work_tattn(args, kw, out):
    c = kw["heads"] * kw["head_dim"]
    return ("temporal_attention", 4.0 * kw["n_pixels"] * kw["frames"] * kw["frames"] * c,
            _numel_bytes(args[0], args[1], out) + kw["n_pixels"] * c * kw["frames"] * 2,
            f"px{kw['n_pixels']} F{kw['frames']} h{kw['heads']} d{kw['head_dim']}")


def _work_mattn(args, kw, out):
    x, w = args[0], args[3]
    rows, c = x.shape
    op = kw.get("out_proj")          # + to_out and the residual (x is read a second time) in the same launch
    return ("motion_attn", 2.0 * rows * c * (4 if op else 3) * c + 4.0 * rows * kw["frames"] * c,
            _numel_bytes(x, w, out) + (_numel_bytes(x, op[0]) if op else 0),
            f"{rows}x{c} F{kw['frames']} h{kw['heads']} d{kw['head_dim']} (LN + q,k,v + attention" + (" + to_out + residual)" if op else ")"))


def _work_cattn(args, kw, out):
    x, w = args[0], args[3]
    rows, c = x.shape
    op = kw.get("out_proj")
    return ("cross_attn_fused", 2.0 * rows * c * (2 if op else 1) * c + 4.0 * rows * kw["ctx_len"] * c,
            _numel_bytes(x, w, args[4], out) + (_numel_bytes(x, op[0]) if op else 0),
            f"{rows}x{c} Lk{kw['ctx_len']} h{kw['heads']} d{kw['head_dim']} (LN + q + text attention" + (" + to_out + residual)" if op else ")"))


def _work_ff(args, kw, out):
    x = args[0]
    rows, c = x.shape
    inner = args[3][1].numel() // 2
    tail = kw.get("tail")
    if tail is not None:       # the block's closing Linear (proj_out + its residual) in the same launch
        return ("ff_fused", 2.0 * rows * c * 2 * inner + 2.0 * rows * inner * c + 2.0 * rows * c * c,
                _numel_bytes(x, args[3][0], args[3][2], tail[0][0], tail[1], out),
                f"{rows}x{c} inner {inner} (LN + GEGLU FF + residual + proj_out + residual{', rows permuted' if tail[2] else ''})")
    return ("ff_fused", 2.0 * rows * c * 2 * inner + 2.0 * rows * inner * c, _numel_bytes(x, args[3][0], args[3][2], out),
            f"{rows}x{c} inner {inner} (LN + GEGLU FF + residual)")


def _work_lnqkv(args, kw, out):
    x, w = args[0], args[3]
    rows, c = x.shape
    return ("ln_qkv", 2.0 * rows * c * (kw["n_qk"] + c), _numel_bytes(x, w, out[0], out[1]),
            f"{rows}x{kw['n_qk']}+{c}x{c} (LN + q,k[,q_adapter] + V^T)")


def _work_gn(args, kw, out):
    if kw.get("stats") is not None:   # statistics from the producer's epilogue: one read, one write
        return "groupnorm", 0.0, 2 * _numel_bytes(out), "x".join(map(str, out.shape)) + " (stats given)"
    return "groupnorm", 0.0, 3 * _numel_bytes(out), "x".join(map(str, out.shape))


def _work_gn_fold(args, kw, out):
    x = args[0]   # statistics pass: one read of x (+ x2); the weight stack is small beside it
    return "groupnorm", 0.0, _numel_bytes(x, kw.get("x2")), "fold " + "x".join(map(str, x.shape))


def _work_ln(args, kw, out):
    return "layernorm", 0.0, 2 * _numel_bytes(out), "x".join(map(str, out.shape))


def _work_misc(name):
    def f(args, kw, out):
        return name, 0.0, 2 * _numel_bytes(out), ""
    return f


_WRAPPED = {
    "gemm": _work_gemm, "conv3x3": _work_conv, "attention": _work_attn, "temporal_attention": _work_tattn,
    "motion_attn": _work_mattn, "cross_attn_fused": _work_cattn, "ff_fused": _work_ff, "ln_qkv": _work_lnqkv,
    "groupnorm": _work_gn, "groupnorm_fold": _work_gn_fold, "layernorm": _work_ln, "silu": _work_misc("elementwise"),
    "copy3d": _work_misc("elementwise"), "timestep_embedding": _work_misc("elementwise"),
    "ddim_prep": _work_misc("elementwise"), "ddim_cfg_step": _work_misc("elementwise"),
    "nchw_to_tokens": _work_misc("elementwise"), "tokens_to_nchw": _work_misc("elementwise"),
}


class KernelProfile:
    def __init__(self):
        self.records = []   # (class, flops, bytes, start event, end event)
        self._saved = {}
        self._depth = 0

    def _wrap(self, name, fn, work):
        def wrapped(*args, **kw):
            if self._depth or kw.get("query_ln_support"):   # nested wrapper: time the outer call only; a support
                return fn(*args, **kw)                      # query launches nothing
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._depth += 1
            s.record()
            try:
                out = fn(*args, **kw)
            finally:
                self._depth -= 1
            e.record()
            cls, flops, nbytes, detail = work(args, kw, out)
            self.records.append((cls, flops, nbytes, s, e, f"{name} {detail}".strip()))
            return out
        return wrapped

    def __enter__(self):
        for name, work in _WRAPPED.items():
            self._saved[name] = getattr(K, name)
            setattr(K, name, self._wrap(name, self._saved[name], work))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(K, name, fn)
        self._saved = {}

    def summary(self):
        """{class: dict(calls, ms, flops, bytes, tflops, gbps)} after a device synchronise."""
        return self._aggregate(lambda rec: rec[0])

    def by_shape(self):
        """the same totals per (wrapper, problem shape, fused extras): which shapes the step's time sits in."""
        return self._aggregate(lambda rec: rec[5])

    def _aggregate(self, key):
        torch.cuda.synchronize()
        agg = {}
        for rec in self.records:
            cls, flops, nbytes, s, e = rec[:5]
            d = agg.setdefault(key(rec), dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        for d in agg.values():
            sec = max(d["ms"], 1e-9) * 1e-3
            d["tflops"] = d["flops"] / sec / 1e12
            d["gbps"] = d["bytes"] / sec / 1e9
        return agg

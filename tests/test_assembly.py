"""Independent check of the UNet ASSEMBLY (channel bookkeeping, block order, IP key order) of both the oracle and the
HIP product against a table written out by hand from the reference's constructor text -- not computed by code that
mirrors those constructors.  (The oracle and the product were written by the same author with near-identical assembly
code, so an assembly mistake would be common-mode and invisible to oracle-vs-product parity tests)

Sources of the literal table (paths relative to /root/reference, SD-1.5 config: block_out_channels (320, 640, 1280,
1280), layers_per_block 2, 8 heads, cross_attention_dim 768):
  * down path channels           src/models/unet_motion_cross_frame_attn.py:783-807 + :200-214 (in = previous out)
  * mid block                    :810-822, :562-619 (resnet, then [attention, motion module, resnet] x 1)
  * up path channels             :831-866 + :380-385: resnet_in = prev_output_channel (i = 0) else out,
                                 skip = in_channels (last layer) else out; in_channels = reversed[min(i + 1, 3)]
  * samplers                     :250-259 (all but the last down block), :431-432 (all but the last up block)
  * attention geometry           :216-231, :397-412: heads = 8, head_dim = C / 8; attn2 keys from 768-d context
  * motion modules               :232-244, :413-425, :607-619: one per resnet (mid: one), C = block width
  * IP-Adapter key ids           :1258-1279: non-motion attn2 layers in attn_processors order, ids 1, 3, 5, ...
  * call order inside blocks     :312-326 (resnet, attention, motion), :509-523 (cat skip, resnet, attention, motion),
                                 :639, :678-692 (resnet 0; attention, motion, resnet 1)
SURVEY.md Appendix B carries the same numbers.
"""
import pytest
import torch

from tests.parity import SD15, SMALL_UNET

TEMB = 1280
CTX = 768
# resnet name -> (input channels incl. the skip concat, output channels)
RESNETS = {
    "down_blocks.0.resnets.0": (320, 320), "down_blocks.0.resnets.1": (320, 320),
    "down_blocks.1.resnets.0": (320, 640), "down_blocks.1.resnets.1": (640, 640),
    "down_blocks.2.resnets.0": (640, 1280), "down_blocks.2.resnets.1": (1280, 1280),
    "down_blocks.3.resnets.0": (1280, 1280), "down_blocks.3.resnets.1": (1280, 1280),
    "mid_block.resnets.0": (1280, 1280), "mid_block.resnets.1": (1280, 1280),
    "up_blocks.0.resnets.0": (2560, 1280), "up_blocks.0.resnets.1": (2560, 1280), "up_blocks.0.resnets.2": (2560, 1280),
    "up_blocks.1.resnets.0": (2560, 1280), "up_blocks.1.resnets.1": (2560, 1280), "up_blocks.1.resnets.2": (1920, 1280),
    "up_blocks.2.resnets.0": (1920, 640), "up_blocks.2.resnets.1": (1280, 640), "up_blocks.2.resnets.2": (960, 640),
    "up_blocks.3.resnets.0": (960, 320), "up_blocks.3.resnets.1": (640, 320), "up_blocks.3.resnets.2": (640, 320),
}
# spatial transformers (with the adapter) and their width; listed in IP-Adapter key-id order (ids 1, 3, ..., 31)
T2D_IN_IP_ORDER = [
    ("down_blocks.0.attentions.0", 320), ("down_blocks.0.attentions.1", 320),
    ("down_blocks.1.attentions.0", 640), ("down_blocks.1.attentions.1", 640),
    ("down_blocks.2.attentions.0", 1280), ("down_blocks.2.attentions.1", 1280),
    ("up_blocks.1.attentions.0", 1280), ("up_blocks.1.attentions.1", 1280), ("up_blocks.1.attentions.2", 1280),
    ("up_blocks.2.attentions.0", 640), ("up_blocks.2.attentions.1", 640), ("up_blocks.2.attentions.2", 640),
    ("up_blocks.3.attentions.0", 320), ("up_blocks.3.attentions.1", 320), ("up_blocks.3.attentions.2", 320),
    ("mid_block.attentions.0", 1280),
]
MOTION = {
    "down_blocks.0": (320, 2), "down_blocks.1": (640, 2), "down_blocks.2": (1280, 2), "down_blocks.3": (1280, 2),
    "mid_block": (1280, 1),
    "up_blocks.0": (1280, 3), "up_blocks.1": (1280, 3), "up_blocks.2": (640, 3), "up_blocks.3": (320, 3),
}
DOWNSAMPLERS = {"down_blocks.0": 320, "down_blocks.1": 640, "down_blocks.2": 1280}      # none on down_blocks.3
UPSAMPLERS = {"up_blocks.0": 1280, "up_blocks.1": 1280, "up_blocks.2": 640}             # none on up_blocks.3
N_PARAMS_SD15 = None   # filled by the first model, the second must agree


def _oracle_cls():
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel
    return UNetMotionCrossFrameAttnModel


def _hip_cls():
    import i2v_adapter_unofficial_amd as p
    return p.UNetMotionCrossFrameAttnModel


@pytest.mark.parametrize("which", ["oracle", "hip"])
def test_sd15_assembly_matches_reference_table(which):
    cls = _oracle_cls() if which == "oracle" else _hip_cls()
    with torch.device("meta"):
        m = cls(**SD15)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    seen = set()

    def expect(key, shape):
        assert key in sd, f"{which}: missing {key}"
        assert sd[key] == tuple(shape), f"{which}: {key} is {sd[key]}, reference layout says {tuple(shape)}"
        seen.add(key)

    expect("conv_in.weight", (320, 4, 3, 3)); expect("conv_in.bias", (320,))
    expect("time_embedding.linear_1.weight", (TEMB, 320)); expect("time_embedding.linear_1.bias", (TEMB,))
    expect("time_embedding.linear_2.weight", (TEMB, TEMB)); expect("time_embedding.linear_2.bias", (TEMB,))
    expect("conv_norm_out.weight", (320,)); expect("conv_norm_out.bias", (320,))
    expect("conv_out.weight", (4, 320, 3, 3)); expect("conv_out.bias", (4,))
    for name, (ci, co) in RESNETS.items():
        expect(f"{name}.norm1.weight", (ci,)); expect(f"{name}.norm1.bias", (ci,))
        expect(f"{name}.conv1.weight", (co, ci, 3, 3)); expect(f"{name}.conv1.bias", (co,))
        expect(f"{name}.time_emb_proj.weight", (co, TEMB)); expect(f"{name}.time_emb_proj.bias", (co,))
        expect(f"{name}.norm2.weight", (co,)); expect(f"{name}.norm2.bias", (co,))
        expect(f"{name}.conv2.weight", (co, co, 3, 3)); expect(f"{name}.conv2.bias", (co,))
        if ci != co:
            expect(f"{name}.conv_shortcut.weight", (co, ci, 1, 1)); expect(f"{name}.conv_shortcut.bias", (co,))
        else:
            assert f"{name}.conv_shortcut.weight" not in sd, f"{which}: unexpected shortcut on {name}"
    modules = dict(m.named_modules())
    for name, c in T2D_IN_IP_ORDER:
        expect(f"{name}.norm.weight", (c,)); expect(f"{name}.norm.bias", (c,))
        expect(f"{name}.proj_in.weight", (c, c, 1, 1)); expect(f"{name}.proj_in.bias", (c,))
        expect(f"{name}.proj_out.weight", (c, c, 1, 1)); expect(f"{name}.proj_out.bias", (c,))
        tb = f"{name}.transformer_blocks.0"
        for n in ("norm1", "norm2", "norm3"):
            expect(f"{tb}.{n}.weight", (c,)); expect(f"{tb}.{n}.bias", (c,))
        for attn, kdim in (("attn1", c), ("attn2", CTX), ("i2v_adapter", c)):
            expect(f"{tb}.{attn}.to_q.weight", (c, c))
            expect(f"{tb}.{attn}.to_k.weight", (c, kdim)); expect(f"{tb}.{attn}.to_v.weight", (c, kdim))
            expect(f"{tb}.{attn}.to_out.0.weight", (c, c)); expect(f"{tb}.{attn}.to_out.0.bias", (c,))
            a = modules[f"{tb}.{attn}"]
            assert (a.heads, a.dim_head) == (8, c // 8), f"{which}: {tb}.{attn} heads/dim_head {a.heads}/{a.dim_head}"
        expect(f"{tb}.ff.net.0.proj.weight", (8 * c, c)); expect(f"{tb}.ff.net.0.proj.bias", (8 * c,))
        expect(f"{tb}.ff.net.2.weight", (c, 4 * c)); expect(f"{tb}.ff.net.2.bias", (c,))
    for blk, (c, n) in MOTION.items():
        for j in range(n):
            name = f"{blk}.motion_modules.{j}"
            expect(f"{name}.norm.weight", (c,)); expect(f"{name}.norm.bias", (c,))
            expect(f"{name}.proj_in.weight", (c, c)); expect(f"{name}.proj_in.bias", (c,))
            expect(f"{name}.proj_out.weight", (c, c)); expect(f"{name}.proj_out.bias", (c,))
            tb = f"{name}.transformer_blocks.0"
            expect(f"{tb}.pos_embed.pe", (1, 32, c))
            for nn_ in ("norm1", "norm2", "norm3"):
                expect(f"{tb}.{nn_}.weight", (c,)); expect(f"{tb}.{nn_}.bias", (c,))
            for attn in ("attn1", "attn2"):                      # double SELF attention: keys from the C-wide tokens
                for w in ("to_q", "to_k", "to_v"):
                    expect(f"{tb}.{attn}.{w}.weight", (c, c))
                expect(f"{tb}.{attn}.to_out.0.weight", (c, c)); expect(f"{tb}.{attn}.to_out.0.bias", (c,))
                a = modules[f"{tb}.{attn}"]
                assert (a.heads, a.dim_head) == (8, c // 8)
            expect(f"{tb}.ff.net.0.proj.weight", (8 * c, c)); expect(f"{tb}.ff.net.0.proj.bias", (8 * c,))
            expect(f"{tb}.ff.net.2.weight", (c, 4 * c)); expect(f"{tb}.ff.net.2.bias", (c,))
        assert f"{blk}.motion_modules.{n}.norm.weight" not in sd
    for blk, c in DOWNSAMPLERS.items():
        expect(f"{blk}.downsamplers.0.conv.weight", (c, c, 3, 3)); expect(f"{blk}.downsamplers.0.conv.bias", (c,))
    for blk, c in UPSAMPLERS.items():
        expect(f"{blk}.upsamplers.0.conv.weight", (c, c, 3, 3)); expect(f"{blk}.upsamplers.0.conv.bias", (c,))
    # the table is exhaustive: nothing else may exist (e.g. attention in down_blocks.3 / up_blocks.0, extra samplers)
    extra = sorted(set(sd) - seen)
    assert not extra, f"{which}: keys outside the reference layout: {extra[:6]} (+{max(0, len(extra) - 6)})"
    n_params = sum(torch.Size(s).numel() for k, s in sd.items() if not k.endswith("pos_embed.pe"))
    assert 1.35e9 < n_params < 1.42e9, n_params        # ~0.86 B UNet + ~0.45 B motion + ~0.05 B adapter
    # IP-Adapter key ids follow attn_processors order (unet:1258-1279)
    ip_names = [n for n in m.attn_processor_names() if n.endswith("attn2.processor") and "motion_modules" not in n]
    assert ip_names == [f"{name}.transformer_blocks.0.attn2.processor" for name, _ in T2D_IN_IP_ORDER]


def test_oracle_block_call_order():
    """call order inside the blocks and the skip wiring of the oracle (the product is compared with the oracle
    numerically on randomised weights, which pins its order to this one)."""
    from tests.parity import small_unet_inputs
    torch.manual_seed(0)
    m = _oracle_cls()(**SMALL_UNET).eval()
    calls = []
    for name, mod in m.named_modules():
        leaf = name.split(".")
        if len(leaf) >= 2 and leaf[-2] in ("resnets", "attentions", "motion_modules", "downsamplers", "upsamplers"):
            mod.register_forward_hook(
                lambda mod_, args, out, name=name: calls.append((name, args[0].shape[1] if args[0].dim() == 4 else None)))
    inp = small_unet_inputs(f=2, hw=8)
    with torch.no_grad():
        m(inp["sample"], inp["timestep"], True, inp["ctx"])
    order = [c[0] for c in calls]
    expected = []
    for i in range(3):
        for j in range(2):
            expected += [f"down_blocks.{i}.resnets.{j}", f"down_blocks.{i}.attentions.{j}", f"down_blocks.{i}.motion_modules.{j}"]
        expected.append(f"down_blocks.{i}.downsamplers.0")
    for j in range(2):
        expected += [f"down_blocks.3.resnets.{j}", f"down_blocks.3.motion_modules.{j}"]
    expected += ["mid_block.resnets.0", "mid_block.attentions.0", "mid_block.motion_modules.0", "mid_block.resnets.1"]
    for j in range(3):
        expected += [f"up_blocks.0.resnets.{j}", f"up_blocks.0.motion_modules.{j}"]
    expected.append("up_blocks.0.upsamplers.0")
    for i in (1, 2, 3):
        for j in range(3):
            expected += [f"up_blocks.{i}.resnets.{j}", f"up_blocks.{i}.attentions.{j}", f"up_blocks.{i}.motion_modules.{j}"]
        if i < 3:
            expected.append(f"up_blocks.{i}.upsamplers.0")
    assert order == expected
    # skip wiring: input width of every up resnet = running width + width of the popped skip (small widths 32/64/128/128)
    widths = {n: c for n, c in calls if ".resnets." in n and n.startswith("up_blocks")}
    assert [widths[f"up_blocks.{i}.resnets.{j}"] for i in range(4) for j in range(3)] == \
        [256, 256, 256, 256, 256, 192, 192, 128, 96, 96, 64, 64]

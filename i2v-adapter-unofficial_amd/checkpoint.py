"""Checkpoint I/O of the weight containers, in the on-disk layout the reference reads and writes, and synthetic
weight initialisation for benchmarks / tests.

The reference's containers are diffusers `ModelMixin` + `ConfigMixin` classes (`I2VAdapterModule` i2v:49-58,
`MotionAdapter`, `UNet2DConditionModel`, `UNetMotionCrossFrameAttnModel` unet:696-730): `save_pretrained(dir)` writes
`config.json` (constructor kwargs + `_class_name`) and `diffusion_pytorch_model.safetensors` (state dict by key);
`from_pretrained(dir)` rebuilds the module from the config and loads the weights (call sites: pipe:733-746,
unet:1080-1116).  `PretrainedMixin` gives the host-side mirrors the same two methods over the same files, so that
checkpoints written by the reference load here by key and vice versa.  No arithmetic happens here.
"""
import inspect
import json
import math
import os
from typing import Optional

import torch
from torch import nn

WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"


def _add_variant(name: str, variant: Optional[str]) -> str:
    if variant is None:
        return name
    stem, ext = name.rsplit(".", 1)
    return f"{stem}.{variant}.{ext}"


def _jsonable(v):
    if isinstance(v, (tuple, list)):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    return v


class PretrainedMixin:
    """`save_pretrained` / `from_pretrained` over diffusers' file layout.  The class must keep its constructor kwargs
    in `self.config` (a dict)."""

    config_name = CONFIG_NAME

    def save_pretrained(self, save_directory: str, is_main_process: bool = True, safe_serialization: bool = True,
                        variant: Optional[str] = None, push_to_hub: bool = False, **_unused):
        if push_to_hub:
            raise NotImplementedError("there is no hub in this build (offline)")
        if os.path.isfile(save_directory):
            raise ValueError(f"Provided path ({save_directory}) should be a directory, not a file")
        if not is_main_process:
            return
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": type(self).__name__}
        cfg.update({k: _jsonable(v) for k, v in dict(self.config).items() if not k.startswith("_")})
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        state = {k: v.detach().to("cpu").contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(state, os.path.join(save_directory, _add_variant(SAFETENSORS_WEIGHTS_NAME, variant)),
                      metadata={"format": "pt"})
        else:
            torch.save(state, os.path.join(save_directory, _add_variant(WEIGHTS_NAME, variant)))

    @classmethod
    def load_config(cls, pretrained_model_path: str, subfolder: Optional[str] = None) -> dict:
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        cfg_file = os.path.join(path, CONFIG_NAME)
        if not os.path.isfile(cfg_file):
            raise EnvironmentError(f"Error no file named {CONFIG_NAME} found in directory {path}.")
        with open(cfg_file) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config):
        keys = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in dict(config).items() if k in keys})

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None,
                        torch_dtype: Optional[torch.dtype] = None, variant: Optional[str] = None,
                        use_safetensors: Optional[bool] = None, **_unused):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        config = cls.load_config(path)
        model = cls.from_config(config)
        st_file = os.path.join(path, _add_variant(SAFETENSORS_WEIGHTS_NAME, variant))
        bin_file = os.path.join(path, _add_variant(WEIGHTS_NAME, variant))
        if use_safetensors is not False and os.path.isfile(st_file):
            from safetensors.torch import load_file
            state = load_file(st_file)
        elif use_safetensors is not True and os.path.isfile(bin_file):
            state = torch.load(bin_file, map_location="cpu", weights_only=True)
        else:
            raise EnvironmentError(f"Error no file named {_add_variant(SAFETENSORS_WEIGHTS_NAME, variant)} or "
                                   f"{_add_variant(WEIGHTS_NAME, variant)} found in directory {path}.")
        state = convert_deprecated_attention_keys(state, model.state_dict())
        missing, unexpected = model.load_state_dict(state, strict=False)
        # recomputable buffers (the sinusoidal table) may be absent from third-party files; anything else is an error
        missing = [k for k in missing if not k.endswith("pos_embed.pe")]
        if missing or unexpected:
            raise RuntimeError(f"{cls.__name__}.from_pretrained({path}): missing keys {missing[:8]} "
                               f"unexpected keys {list(unexpected)[:8]}")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()


_DEPRECATED_ATTENTION_NAMES = ((".query.", ".to_q."), (".key.", ".to_k."), (".value.", ".to_v."),
                               (".proj_attn.", ".to_out.0."))


def convert_deprecated_attention_keys(state: dict, target: dict) -> dict:
    """diffusers converts the attention names of old checkpoints on load (`_convert_deprecated_attention_blocks`, reached
    from `AutoencoderKL.from_pretrained`, pipe:754): many SD-1.5 `vae/` folders (runwayml/stable-diffusion-v1-5,
    sd-vae-ft-mse) still store `mid_block.attentions.0.{query,key,value,proj_attn}`, which the current module tree calls
    `{to_q,to_k,to_v,to_out.0}`.  Keys the target already knows are left alone; a 1x1-conv weight [C, C, 1, 1] is
    squeezed where the target parameter is a matrix."""
    out = {}
    for k, v in state.items():
        nk = k
        if k not in target and ".attentions." in k:
            for old, new in _DEPRECATED_ATTENTION_NAMES:
                if old in nk:
                    nk = nk.replace(old, new)
        if nk in target and v.dim() == 4 and target[nk].dim() == 2 and v.shape[2:] == (1, 1):
            v = v[:, :, 0, 0]
        out[nk] = v
    return out


def load_ip_adapter_file(pretrained_model_name_or_path_or_dict, subfolder: Optional[str] = None,
                         weight_name: Optional[str] = None) -> dict:
    """`ip-adapter_sd15.bin` / `.safetensors` -> {"image_proj": {...}, "ip_adapter": {...}} (the argument of
    `_load_ip_adapter_weights`, unet:1230-1239; file handling of diffusers' IPAdapterMixin.load_ip_adapter,
    called at pipe:783)."""
    if isinstance(pretrained_model_name_or_path_or_dict, dict):
        return pretrained_model_name_or_path_or_dict
    path = pretrained_model_name_or_path_or_dict
    if subfolder:
        path = os.path.join(path, subfolder)
    if weight_name:
        path = os.path.join(path, weight_name)
    if not os.path.isfile(path):
        raise EnvironmentError(f"IP-Adapter weights not found at {path}")
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        state = {"image_proj": {}, "ip_adapter": {}}
        with safe_open(path, framework="pt", device="cpu") as f:
            for key in f.keys():
                if key.startswith("image_proj."):
                    state["image_proj"][key[len("image_proj."):]] = f.get_tensor(key)
                elif key.startswith("ip_adapter."):
                    state["ip_adapter"][key[len("ip_adapter."):]] = f.get_tensor(key)
    else:
        state = torch.load(path, map_location="cpu", weights_only=True)
    if list(state.keys()) != ["image_proj", "ip_adapter"] and set(state.keys()) != {"image_proj", "ip_adapter"}:
        raise ValueError("Required keys are (`image_proj` and `ip_adapter`) missing from the state dict.")
    return state


@torch.no_grad()
def init_random_weights_(model: nn.Module, seed: int = 1234, adapter_out_std: float = 0.02,
                         norm_jitter: float = 0.0) -> nn.Module:
    """Synthetic weights for benchmarks and parity tests (there are no pretrained files offline), drawn on the
    model's own device (works on modules materialised with `to_empty`): torch's default Linear / Conv2d law
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norms = 1 / 0 (+ `norm_jitter` * N(0,1) so that tests see the affine
    parameters); sinusoidal tables recomputed; every adapter `to_out` ~ N(0, adapter_out_std^2) (a
    freshly assembled model has zero adapter to_out, i2v:181-182, and K1 would not contribute)."""
    from .blocks import SinusoidalPositionalEmbedding
    p0 = next(model.parameters())
    g = torch.Generator(device=p0.device).manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            bound = 1.0 / math.sqrt(m.weight[0].numel())
            m.weight.uniform_(-bound, bound, generator=g)
            if m.bias is not None:
                m.bias.uniform_(-bound, bound, generator=g)
        elif isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
            if m.weight is not None:
                m.weight.fill_(1.0)
                m.bias.zero_()
                if norm_jitter:
                    m.weight.add_(torch.randn(m.weight.shape, generator=g, device=p0.device,
                                              dtype=torch.float32).to(m.weight.dtype) * norm_jitter)
                    m.bias.add_(torch.randn(m.bias.shape, generator=g, device=p0.device,
                                            dtype=torch.float32).to(m.bias.dtype) * norm_jitter)
        elif isinstance(m, SinusoidalPositionalEmbedding):
            m.reset_table_()
    for name, p in model.named_parameters():
        if ".i2v_adapter.to_out." in name or name.startswith("i2v_adapter.to_out."):
            p.copy_((torch.randn(p.shape, generator=g, device=p0.device, dtype=torch.float32) * adapter_out_std)
                    .to(p.dtype))
    return model

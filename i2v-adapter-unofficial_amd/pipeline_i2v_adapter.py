"""I2VAdapterPipeline denoising loop on the HIP kernels: drop-in for the hot part of
/root/reference/src/pipelines/pipeline_i2v_adapter.py (`__init__` pipe:75-112, `prepare_latents` :265-297,
`get_timesteps` :529-536, `__call__` :538-719, loop :663-700).

One DDIM step = frame-0 overwrite + CFG duplicate + layout edge (i2v_ddim_prep), timestep embedding, the UNet,
CFG combine + DDIM update (i2v_ddim_cfg_step).  The whole step is captured ONCE as a hipGraph (through
torch.cuda.CUDAGraph on the stream the ctypes launches go to) and replayed for every timestep: the step's
scalars (t, sqrt(a_t), ...) are read on the device from small tables indexed by a device-side step counter, so
there is no per-step host<->device traffic and no per-step sync (the reference syncs once per step on
`alphas_cumprod[t]`).

Either side of the loop: the condition image is encoded and the final latents are decoded by the HIP
AutoencoderKL (vae.py: pipe:300-320, 626-627) when a `vae` is given; pre / post-processing, `tensor2vid` and GIF
export are host-side plumbing (image_processor.py).  Out of scope: the CLIP text / image
encoders -- pass `prompt_embeds`, `negative_prompt_embeds` and optional `image_embeds`.
"""
import os
from typing import Optional

import torch

from . import kernels as K
from ._lib import HipLibraryError
from .blocks import DDIMScheduler
from .image_processor import VaeImageProcessor, tensor2vid
from .unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel

# The two classifier-free-guidance halves of a step's batch are the same latents at the same timestep until the first text
# cross-attention: that prefix of the UNet is computed once (same numbers bit for bit; I2V_CFG_SHARED=0 computes it twice,
# as the reference does).
CFG_SHARED = os.environ.get("I2V_CFG_SHARED", "1") != "0"

f16 = torch.float16


class I2VAdapterPipelineOutput:
    def __init__(self, frames):
        self.frames = frames


def draw_blur_sigma(generator=None, sigma_min: float = 0.1, sigma_max: float = 2.0) -> float:
    """torchvision GaussianBlur(kernel_size=3) (pipe:112) draws sigma ~ U(0.1, 2.0) once per call
    (`GaussianBlur.get_params`); the reference uses the unseeded global RNG, here the draw takes a generator."""
    gdev = generator.device if generator is not None else torch.device("cpu")
    return float(torch.empty(1, device=gdev).uniform_(sigma_min, sigma_max, generator=generator).item())


def _draw(fn, shape, generator, device):
    """one seeded draw on the generator's own device (host generators reproduce the CPU oracle's numbers bit for bit;
    a torch.Generator(device="cuda") keeps the whole prior on the GPU), moved to `device`.  A list of generators draws
    one sample (leading index) per generator, as `prepare_latents` does (pipe:287-290): a sample's noise then does not
    depend on which other samples share its call."""
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"{len(generator)} generators for a batch of {shape[0]}")
        return torch.cat([_draw(fn, (1,) + tuple(shape[1:]), g, device) for g in generator], dim=0)
    gdev = generator.device if generator is not None else torch.device("cpu")
    return fn(shape, generator=generator, dtype=torch.float32, device=gdev).to(device)


class I2VAdapterPipeline:
    """Reference constructor order (pipe:75-93): (vae, text_encoder, tokenizer, unet, motion_adapter, i2v_adapter,
    scheduler, feature_extractor, image_encoder).  `unet` may be an SD-1.5-layout `UNet2DConditionModel` container
    (then the motion UNet is assembled with `from_unet2d`, pipe:96) or an already built
    `UNetMotionCrossFrameAttnModel`."""

    vae_scale_factor = 8
    model_cpu_offload_seq = "text_encoder->image_encoder->unet->vae"

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, motion_adapter=None,
                 i2v_adapter=None, scheduler: Optional[DDIMScheduler] = None, feature_extractor=None,
                 image_encoder=None):
        if unet is None:
            raise ValueError("`unet` is required")
        if not isinstance(unet, UNetMotionCrossFrameAttnModel):
            unet = UNetMotionCrossFrameAttnModel.from_unet2d(unet, motion_adapter, i2v_adapter)     # pipe:96
        self.unet = unet
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.motion_adapter, self.i2v_adapter = motion_adapter, i2v_adapter
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler()
        self.feature_extractor, self.image_encoder = feature_extractor, image_encoder
        if vae is not None:                                                                  # pipe:110-111
            self.vae_scale_factor = 2 ** (len(vae.config["block_out_channels"]) - 1)
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)
        self._vae_slicing = False
        self._graph = None
        self._graph_cache = {}

    def enable_vae_slicing(self):
        """pipe:122-128: decode the frames one at a time (peak activation memory / num_frames)."""
        self._vae_slicing = True

    def disable_vae_slicing(self):
        self._vae_slicing = False

    def decode_latents(self, latents):
        """pipe:300-320: latents (B, F, 4, h, w) -> video (B, F, 3, 8h, 8w) float32 through the HIP VAE decoder."""
        if self.vae is None:
            raise ValueError("decode_latents needs a `vae` (AutoencoderKL) -- or use output_type='latent'")
        latents = 1 / self.vae.config["scaling_factor"] * latents
        b, f, c, h, w = latents.shape
        flat = latents.reshape(b * f, c, h, w)
        if self._vae_slicing:
            image = torch.cat([self.vae.decode(flat[i: i + 1]).sample for i in range(b * f)])
        else:
            image = self.vae.decode(flat).sample
        return image[None, :].reshape((b, f, -1) + image.shape[2:]).float()

    def encode_condition_image(self, condition_image, height, width, generator=None):
        """pipe:626-627: preprocess -> vae.encode(...).latent_dist.sample() * scaling_factor."""
        if self.vae is None:
            raise ValueError("a `condition_image` needs a `vae` (AutoencoderKL) -- or pass `condition_image_latents`")
        img = self.image_processor.preprocess(condition_image, height=height, width=width).to(self.vae.device)
        if isinstance(generator, list):
            generator = generator[0]
        return self.vae.encode(img).latent_dist.sample(generator) * self.vae.config["scaling_factor"]

    def to(self, device=None, dtype=None):
        """move the models the pipeline holds (pipe:784); fp16 is the storage dtype of the HIP path."""
        for name in ("unet", "vae"):
            m = getattr(self, name)
            if m is not None:
                setattr(self, name, m.to(device=device, dtype=dtype))
        return self

    def load_i2v_adapter(self, i2v_adapter):
        self.unet.load_i2v_adapter(i2v_adapter)
        self.i2v_adapter = i2v_adapter

    def load_ip_adapter(self, pretrained_model_name_or_path_or_dict, subfolder: Optional[str] = None,
                        weight_name: Optional[str] = None, **_unused):
        """diffusers IPAdapterMixin.load_ip_adapter as called at pipe:783: reads ip-adapter_sd15.{bin,safetensors} and
        installs the decoupled cross-attention branch + ImageProjection (unet:1230-1287).  (The CLIP image encoder it
        also loads upstream is out of scope: pass `image_embeds`.)"""
        from .checkpoint import load_ip_adapter_file
        self.unet._load_ip_adapter_weights(
            load_ip_adapter_file(pretrained_model_name_or_path_or_dict, subfolder=subfolder, weight_name=weight_name))

    def load_motion_adapter(self, motion_adapter):
        self.unet.load_motion_modules(motion_adapter)
        self.motion_adapter = motion_adapter

    def get_timesteps(self, num_inference_steps, strength, device=None):
        """pipe:529-536."""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start:], num_inference_steps - t_start

    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, dtype, device,
                        generator, latents=None):
        """pipe:265-297 (noise drawn on the host generator so that a seed reproduces the CPU oracle's draw)."""
        shape = (batch_size, num_frames, num_channels_latents, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):                                   # one generator per sample (pipe:287-290)
                latents = torch.cat([_draw(torch.randn, (1,) + shape[1:], g, device) for g in generator], dim=0)
            else:
                latents = _draw(torch.randn, shape, generator, device)
        return latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------------------------------ one step
    def _step(self, st):
        """One iteration of pipe:666-697 as kernel launches on the current stream (captured into a hipGraph)."""
        unet = self.unet
        x = K.ddim_prep(st["latents"], st["cond"], unet.packed()["cin_pad"], st["copies"])    # pipe:668-673
        # the time-embedding chain of this step's timestep: one row of the table computed once per sample (_time_table)
        temb_proj = K.select_row(st["temb_table"], st["step_idx"])
        # the CFG halves are copies of one tensor at one timestep (pipe:672-673): what does not depend on the prompt is
        # computed once (unet._fwd_tokens, cfg_shared)
        y = unet._fwd_tokens(x, None, True, st.get("ctx_proj") or st["ctx_text"], st["ctx_ip"],
                             st["num_frames"], cfg_shared=CFG_SHARED and st["copies"] == 2, temb_proj=temb_proj,
                             forward_upsample_size=any(s % (2 ** unet.num_upsamplers) for s in st["latents"].shape[-2:]))   # pipe:676-683, unet:1304-1311
        K.ddim_cfg_step(st["latents"], y, st["coef"], st["step_idx"], st["guidance"], st["copies"])  # pipe:686-691

    def _graph_key(self, st):
        """everything a captured step has baked in besides the contents of the static buffers: shapes, the Python
        scalars passed as launch arguments (guidance, IP scales) and the identity / version of every weight (the packed
        kernel-layout copies are rebuilt when a parameter changes, and a graph captured before that reads the old ones)"""
        unet = self.unet
        wsig = hash(tuple((p.data_ptr(), p._version) for p in unet.parameters()))
        ips = tuple((a.ip_num_tokens, float(a.ip_scale)) for a in unet._cross_attention_layers())
        shp = lambda t: None if t is None else (tuple(t.shape), t.dtype)
        from .blocks import precise_stream      # (a captured step keeps the residual-stream mode it was captured in)
        return (tuple(st["latents"].shape), st["copies"], st["num_frames"], st["guidance"], shp(st["t_table"]),
                shp(st["ctx_text"]), shp(st["ctx_ip"]), str(st["latents"].device), wsig, ips, precise_stream())

    def _run_steps(self, st, n_steps, use_graph):
        if not use_graph:
            st["ctx_proj"] = self.unet.project_context(st["ctx_text"], st["ctx_ip"])
            st["temb_table"] = self.unet.project_time_table(st["t_table"])
            for _ in range(n_steps):
                self._step(st)
            return st["latents"]
        # The captured step is kept across calls: a second sample of the same shape (the evaluation driver's loop) copies
        # its inputs into the graph's static buffers and replays -- no eager warm-up step, no re-capture (~0.3 s of a
        # 1.8 s sample at 16 f x 512 x 512).
        key = self._graph_key(st)
        cache = self._graph_cache
        hit = cache.get(key)
        if hit is not None:
            graph, gst = hit
            for name in ("latents", "cond", "t_table", "coef", "ctx_text", "ctx_ip"):
                if st[name] is not None:
                    gst[name].copy_(st[name])
            gst["step_idx"].zero_()
            self.unet.project_context(gst["ctx_text"], gst["ctx_ip"], out=gst["ctx_proj"])
            self.unet.project_time_table(gst["t_table"], out=gst["temb_table"])
        else:
            # one shape at a time (a graph pins its workspace): the old graph and its pool go before the new capture,
            # so two pools never coexist at the peak
            cache.clear()
            self._graph = None
            st["ctx_proj"] = self.unet.project_context(st["ctx_text"], st["ctx_ip"])
            st["temb_table"] = self.unet.project_time_table(st["t_table"])
            # warm-up outside capture: packs weights, sizes the allocator; then restore the state it advanced
            saved = st["latents"].clone()
            self._step(st)
            st["latents"].copy_(saved)
            st["step_idx"].zero_()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._step(st)
            st["latents"].copy_(saved)      # capture does not execute, but keep the invariant explicit
            st["step_idx"].zero_()
            gst = dict(st)                  # its own dict: the caller goes on to rebind entries of `st`
            cache[key] = (graph, gst)
        self._graph = graph
        for _ in range(n_steps):
            graph.replay()
        return gst["latents"].clone()     # the static buffer stays with the graph; the caller gets its own tensor

    def _resolve_call_defaults(self, output_type, callback, use_graph):
        """(output_type, use_graph) as `__call__` uses them.  output_type=None is the reference's default "pil" (pipe:556);
        a pipeline assembled without a VAE (latents in, latents out: the benchmarks and most tests) has nothing to decode
        with and returns the latents.  pipe:693-697 calls back after every step: a replayed hipGraph has no per-step host
        hook, so a `callback` runs the same kernels as eager launches (bit-identical, tests/test_modules_gpu.py)."""
        if output_type is None:
            output_type = "pil" if self.vae is not None else "latent"
        if callback is not None:
            use_graph = False
        return output_type, use_graph

    # ------------------------------------------------------------------------------------------ __call__
    @torch.no_grad()
    def __call__(self, prompt=None, condition_image=None, num_frames: Optional[int] = 16,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt=None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 ip_adapter_image=None, output_type: Optional[str] = None, return_dict: bool = True,
                 callback=None, callback_steps: Optional[int] = 1, cross_attention_kwargs=None, clip_skip=None,
                 frame_similarity_sample_ratio: float = 1, frame_similarity_blurred_strength: float = 0.6,
                 condition_image_latents=None, image_embeds=None, negative_image_embeds=None,
                 prior_mask_generator=None, prior_noise_generator=None, blur_sigma: Optional[float] = None,
                 use_graph: bool = True, precise_stream: Optional[bool] = None):
        """pipe:537-711 (the arguments the reference's `__call__` takes; `use_graph`, the explicit prior generators and
        `precise_stream` are additions).  precise_stream: True / False runs THIS call with / without the precise residual stream
        (fp16 hi + lo pairs between the UNet's modules, DESIGN 2.2: closer to the fp32 reference, +2.6 % step time); None keeps the
        process setting (`blocks.set_precise_stream`, I2V_STREAM_PRECISE)."""
        if precise_stream is not None:
            kw = dict(locals())
            for k in ("self", "precise_stream"):
                kw.pop(k)
            from . import blocks
            prev = blocks.set_precise_stream(precise_stream)
            try:
                return self.__call__(**kw)
            finally:
                blocks.set_precise_stream(prev)
        if prompt is not None or ip_adapter_image is not None:
            raise NotImplementedError(
                "the CLIP text / image encoders are out of scope of this build (SURVEY section 2 row 3b): pass "
                "`prompt_embeds`, `negative_prompt_embeds` (and `image_embeds`)")
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and "
                             "`prompt_embeds` undefined.")
        if condition_image is not None and condition_image_latents is None:                     # pipe:624-627
            if height is None or width is None:
                height = height or self.unet.config.sample_size * self.vae_scale_factor         # pipe:568-569
                width = width or self.unet.config.sample_size * self.vae_scale_factor
            if height % 8 != 0 or width % 8 != 0:                                               # pipe:213-214
                raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
            condition_image_latents = self.encode_condition_image(condition_image, height, width, generator)
        if condition_image_latents is None:
            raise ValueError("`condition_image` (or `condition_image_latents`) is required: the reference's prior "
                             "(pipe:647-656) needs the condition image and crashes without it")
        output_type, use_graph = self._resolve_call_defaults(output_type, callback, use_graph)
        dev = self.unet.device
        if dev.type != "cuda":
            raise HipLibraryError(f"unet is on {dev}: the HIP path has no CPU fallback")
        h_lat, w_lat = condition_image_latents.shape[-2:]
        height = height or h_lat * self.vae_scale_factor
        width = width or w_lat * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:                                                   # pipe:213-214
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        # (latent sizes that are not multiples of 8 take the UNet's forward_upsample_size path, unet:1304-1311)
        assert 0 < frame_similarity_sample_ratio <= 1, (
            f'"frame_similarity_sample_ratio" for img2vid must in (0, 1]. But receive {frame_similarity_sample_ratio}.')
        batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0
        copies = 2 if do_cfg else 1
        if do_cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed "
                                 f"directly, but got: `prompt_embeds` {prompt_embeds.shape} != "
                                 f"`negative_prompt_embeds` {negative_prompt_embeds.shape}.")
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])                  # pipe:613-614
            if image_embeds is not None:
                if negative_image_embeds is None:
                    negative_image_embeds = torch.zeros_like(image_embeds)                      # pipe:343
                image_embeds = torch.cat([negative_image_embeds, image_embeds])                 # pipe:621-622

        self.scheduler.set_timesteps(num_inference_steps)                                       # pipe:630-631
        timesteps, _ = self.get_timesteps(num_inference_steps, frame_similarity_sample_ratio)

        cond_dev = condition_image_latents.detach().to(dev, torch.float32).contiguous()
        latents = self.prepare_latents(batch_size, self.unet.config.in_channels, num_frames, height, width,
                                       torch.float32, dev, generator, latents)                  # pipe:635-645
        # first-frame-similarity prior + add_noise (pipe:647-656) in ONE HIP kernel; the reference overwrites the
        # latents drawn above (its `latents = self.scheduler.add_noise(...)`, pipe:656), and so does this.  The random
        # draws (blur sigma, mask, noise) take explicit generators (the reference uses the unseeded global RNG).
        if blur_sigma is None:                                                                  # pipe:112
            blur_sigma = draw_blur_sigma(prior_mask_generator[0] if isinstance(prior_mask_generator, (list, tuple))
                                         else prior_mask_generator)
        shape = (batch_size, num_frames) + tuple(cond_dev.shape[1:])
        mask_u = _draw(torch.rand, shape, prior_mask_generator, dev)                            # pipe:652
        noise = _draw(torch.randn, shape, prior_noise_generator, dev)                           # pipe:655
        a_t = float(self.scheduler.alphas_cumprod[int(timesteps[0])])
        latents = K.first_frame_prior(cond_dev, mask_u.contiguous(), noise.contiguous(), blur_sigma,
                                      frame_similarity_blurred_strength, a_t ** 0.5, (1.0 - a_t) ** 0.5)

        st = dict(
            latents=latents, cond=cond_dev, copies=copies,
            num_frames=num_frames, guidance=float(guidance_scale),
            t_table=timesteps.to(torch.float32).to(dev), coef=self.scheduler.step_coefficients(timesteps, eta).to(dev),
            step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
            ctx_text=prompt_embeds.to(dev, f16).contiguous(),
            ctx_ip=self.unet._project_image_embeds(
                {"image_embeds": image_embeds.to(dev)} if image_embeds is not None else None))
        # K / V^T of the prompt (+ image) context for all 16 cross-attention layers: once per sample, not once per step
        # (projected where it is consumed: a graph-cache hit projects straight into the graph's static buffers)
        if callback is None and eta == 0.0:
            st["latents"] = self._run_steps(st, len(timesteps), use_graph)
        else:
            # eager steps: a per-step host hook (pipe:693-697), and / or the stochastic DDIM update (eta > 0, pipe:550, 659-660:
            # sigma_t is out of the direction coefficient -- `step_coefficients(timesteps, eta)` -- and comes back as fresh noise,
            # one draw of the latents' shape per step from `generator` as diffusers' scheduler draws it)
            if eta != 0.0 and use_graph:
                import warnings
                warnings.warn("eta > 0: the stochastic DDIM update draws fresh noise on the host every step, so the steps run as "
                              "eager launches instead of the captured hipGraph (about 2x the step time)", RuntimeWarning, stacklevel=2)
            sigmas = self.scheduler.step_sigmas(timesteps, eta)
            st["ctx_proj"] = self.unet.project_context(st["ctx_text"], st["ctx_ip"])
            st["temb_table"] = self.unet.project_time_table(st["t_table"])
            for i, t in enumerate(timesteps):                                                   # pipe:666-697
                self._step(st)
                if eta > 0:
                    z = _draw(torch.randn, tuple(st["latents"].shape), generator, dev).to(torch.float32).contiguous()
                    K.axpby(st["latents"], z, 1.0, sigmas[i])
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, st["latents"])
        latents = st["latents"]
        latents[:, 0] = st["cond"]                                                              # pipe:699-700
        if output_type == "latent":                                                             # pipe:702-703
            video = latents
        else:
            video_tensor = self.decode_latents(latents)                                         # pipe:706
            video = video_tensor if output_type == "pt" else tensor2vid(video_tensor, self.image_processor,
                                                                        output_type=output_type)  # pipe:708-711
        if not return_dict:
            return (video,)
        return I2VAdapterPipelineOutput(frames=video)


def main(argv=None):
    """The reference's evaluation driver (pipe:721-809, the README command
    `python src/pipelines/pipeline_i2v_adapter.py --task_name ... --checkpoint_epoch ...`): load MotionAdapter /
    I2VAdapterModule / SD-1.5 UNet + VAE / IP-Adapter from the reference's directory layout, read the CSV of
    (image_path, name) pairs, sample 16 frames per pair and write one GIF per prompt.

    The CLIP text / image encoders are out of scope of this build, so the per-row `prompt_embeds`,
    `negative_prompt_embeds` (and `image_embeds`) come from a safetensors file (--embeds) written by
    `src/tools/encode_text.py`-style tooling; everything else follows the reference driver."""
    import argparse
    import logging
    import os

    import pandas as pd
    import PIL.Image
    from safetensors.torch import load_file

    from .blocks import MotionAdapter
    from .i2v_adapter import I2VAdapterModule
    from .image_processor import export_to_gif
    from .unet_motion_cross_frame_attn import UNet2DConditionModel
    from .vae import AutoencoderKL

    logger = logging.getLogger("i2v_adapter_pipeline")
    logging.basicConfig(level=logging.INFO)
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint_epoch", type=int, default=0)
    parser.add_argument("--eval_data_path", type=str, default="./data/WebVid-10M/I2VAdapter-eval.csv")
    parser.add_argument("--task_name", type=str)
    parser.add_argument("--embeds", type=str, required=True,
                        help="safetensors with prompt_embeds [N,77,768], negative_prompt_embeds [N or 1,77,768], "
                             "optional image_embeds [N,1024] (row i = CSV row i)")
    parser.add_argument("--model_path", type=str, default="./SG161222_Realistic_Vision_V5.1_noVAE/")
    parser.add_argument("--motion_adapter_path", type=str, default="./animatediff-motion-adapter-v1-5-2")
    parser.add_argument("--ip_adapter_path", type=str, default="./IP-Adapter/")
    parser.add_argument("--checkpoint_root", type=str, default="./checkpoint")
    parser.add_argument("--samples_root", type=str, default="./samples")
    parser.add_argument("--num_frames", type=int, default=16)
    parser.add_argument("--num_inference_steps", type=int, default=25)
    parser.add_argument("--height", type=int, default=None)
    parser.add_argument("--width", type=int, default=None)
    parser.add_argument("--seed", type=int, default=0)
    args = parser.parse_args(argv)
    if args.task_name is None:
        logger.error("Checkpoint `task_name` must be specified.")
        return -1

    i2v_adapter = None
    motion_adapter = MotionAdapter.from_pretrained(args.motion_adapter_path)                     # pipe:734
    checkpoint_path = os.path.join(args.checkpoint_root, args.task_name, f"epoch_{args.checkpoint_epoch}")
    i2v_adapter_path = os.path.join(checkpoint_path, "i2v_adapter")
    if not os.path.exists(i2v_adapter_path):
        logger.warning(f"Fatal! Checkpoint path {i2v_adapter_path} for I2VAdapterModule doesnot exist!")
    else:
        i2v_adapter = I2VAdapterModule.from_pretrained(i2v_adapter_path)                         # pipe:741
        logger.info(f"Successfully loaded I2VAdapterModule from {i2v_adapter_path}.")
    motion_adapter_path = os.path.join(checkpoint_path, "motion_modules")
    if os.path.exists(motion_adapter_path):
        motion_adapter = MotionAdapter.from_pretrained(motion_adapter_path)                      # pipe:745
        logger.info(f"Successfully loaded MotionModule from {motion_adapter_path}.")

    device = torch.device("cuda")                                # there is no CPU path (the reference falls back to it)
    unet2d = UNet2DConditionModel.from_pretrained(os.path.join(args.model_path, "unet"))         # pipe:751
    vae = AutoencoderKL.from_pretrained(os.path.join(args.model_path, "vae"))                    # pipe:754
    scheduler = DDIMScheduler.from_pretrained(args.model_path, subfolder="scheduler", clip_sample=False,
                                              timestep_spacing="linspace", steps_offset=1)       # pipe:755-757

    eval_data_dir = os.path.dirname(args.eval_data_path)                                         # pipe:759-768
    eval_data_df = pd.read_csv(args.eval_data_path)
    condition_images = [PIL.Image.open(os.path.join(eval_data_dir, p)) for p in eval_data_df["image_path"]]
    eval_prompts = eval_data_df["name"].tolist()
    emb = load_file(args.embeds)
    n = len(eval_prompts)
    if emb["prompt_embeds"].shape[0] != n:
        raise ValueError(f"{args.embeds} holds {emb['prompt_embeds'].shape[0]} prompt embeddings for {n} CSV rows")

    pipe = I2VAdapterPipeline(vae, None, None, unet2d.to(device).half(), motion_adapter, i2v_adapter, scheduler)
    if "image_embeds" in emb:                                                                    # pipe:783
        pipe.load_ip_adapter(args.ip_adapter_path, subfolder="models", weight_name="ip-adapter_sd15.bin")
    pipe.to(device, torch.float16)
    pipe.enable_vae_slicing()                                                                    # pipe:787

    sample_save_dir = os.path.join(args.samples_root, args.task_name, f"epoch_{args.checkpoint_epoch}")
    os.makedirs(sample_save_dir, exist_ok=True)
    neg = emb["negative_prompt_embeds"]
    for ind in range(n):                                         # one sample per call: every sample replays the graph
        g = lambda k: torch.Generator().manual_seed(args.seed * 1000 + 10 * ind + k)
        out = pipe(prompt_embeds=emb["prompt_embeds"][ind: ind + 1],
                   negative_prompt_embeds=neg[ind: ind + 1] if neg.shape[0] == n else neg[:1],
                   image_embeds=emb["image_embeds"][ind: ind + 1] if "image_embeds" in emb else None,
                   condition_image=condition_images[ind], num_frames=args.num_frames, guidance_scale=7.5,
                   num_inference_steps=args.num_inference_steps, frame_similarity_sample_ratio=0.9,
                   height=args.height, width=args.width, output_type="pil", generator=g(0),
                   prior_mask_generator=g(1), prior_noise_generator=g(2))                        # pipe:790-799
        export_to_gif(out.frames[0], os.path.join(sample_save_dir, f"{eval_prompts[ind]}.gif"))  # pipe:806-807
    logger.info(f"Finish sampling {n} instances, the results saved to {sample_save_dir}.")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

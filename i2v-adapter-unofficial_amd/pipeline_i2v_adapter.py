"""I2VAdapterPipeline denoising loop on the HIP kernels: drop-in for the hot part of
/root/reference/src/pipelines/pipeline_i2v_adapter.py (`__init__` pipe:75-112, `prepare_latents` :265-297,
`get_timesteps` :529-536, `__call__` :538-719, loop :663-700).

One DDIM step = frame-0 overwrite + CFG duplicate + layout edge (i2v_ddim_prep), timestep embedding, the UNet,
CFG combine + DDIM update (i2v_ddim_cfg_step).  The whole step is captured ONCE as a hipGraph (through
torch.cuda.CUDAGraph on the stream the ctypes launches go to) and replayed for every timestep: the step's
scalars (t, sqrt(a_t), ...) are read on the device from small tables indexed by a device-side step counter, so
there is no per-step host<->device traffic and no per-step sync (the reference syncs once per step on
`alphas_cumprod[t]`).

Out of scope here: CLIP text / image encoders, VAE encode / decode, PIL / GIF I/O.
The loop takes `prompt_embeds`, `negative_prompt_embeds`, `condition_image_latents` and optional `image_embeds`.
"""
from typing import Optional

import torch

from . import kernels as K
from ._lib import HipLibraryError
from .blocks import DDIMScheduler
from .unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel

f16 = torch.float16


class I2VAdapterPipelineOutput:
    def __init__(self, frames):
        self.frames = frames


def gaussian_blur3(x: torch.Tensor, sigma: float) -> torch.Tensor:
    """torchvision GaussianBlur(kernel_size=3) for one sigma (pipe:112,648): separable 3-tap kernel, reflect pad.
    Runs once per sample on the host, before the loop."""
    xs = torch.linspace(-1.0, 1.0, 3)
    pdf = torch.exp(-0.5 * (xs / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :]).to(x.dtype)
    c = x.shape[-3]
    shp = x.shape
    x4 = torch.nn.functional.pad(x.reshape(-1, c, shp[-2], shp[-1]), (1, 1, 1, 1), mode="reflect")
    return torch.nn.functional.conv2d(x4, k2.expand(c, 1, 3, 3), groups=c).reshape(shp)


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
        self._graph = None
        self._graph_key = None

    def load_i2v_adapter(self, i2v_adapter):
        self.unet.load_i2v_adapter(i2v_adapter)
        self.i2v_adapter = i2v_adapter

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
            latents = torch.randn(shape, generator=generator, dtype=torch.float32)
        return latents.to(torch.float32) * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------------------------------ one step
    def _step(self, st):
        """One iteration of pipe:666-697 as kernel launches on the current stream (captured into a hipGraph)."""
        unet = self.unet
        x = K.ddim_prep(st["latents"], st["cond"], unet.packed()["cin_pad"], st["copies"])    # pipe:668-673
        temb = unet._embed_time(st["t_table"], t_index=st["step_idx"])
        y = unet._fwd_tokens(x, temb, True, st["ctx_text"], st["ctx_ip"], st["num_frames"])   # pipe:676-683
        K.ddim_cfg_step(st["latents"], y, st["coef"], st["step_idx"], st["guidance"], st["copies"])  # pipe:686-691

    def _run_steps(self, st, n_steps, use_graph):
        if not use_graph:
            for _ in range(n_steps):
                self._step(st)
            return
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
        self._graph = graph
        for _ in range(n_steps):
            graph.replay()

    # ------------------------------------------------------------------------------------------ __call__
    @torch.no_grad()
    def __call__(self, prompt=None, condition_image=None, num_frames: Optional[int] = 16,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt=None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 ip_adapter_image=None, output_type: Optional[str] = "latent", return_dict: bool = True,
                 callback=None, callback_steps: Optional[int] = 1, cross_attention_kwargs=None, clip_skip=None,
                 frame_similarity_sample_ratio: float = 1, frame_similarity_blurred_strength: float = 0.6,
                 condition_image_latents=None, image_embeds=None, negative_image_embeds=None,
                 prior_mask_generator=None, prior_noise_generator=None, blur_sigma: float = 1.0,
                 use_graph: bool = True):
        if prompt is not None or condition_image is not None or ip_adapter_image is not None:
            raise NotImplementedError(
                "text / image encoders and the VAE are out of scope of this build (SURVEY section 2 row 3b): pass "
                "`prompt_embeds`, `negative_prompt_embeds`, `condition_image_latents` (and `image_embeds`)")
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and "
                             "`prompt_embeds` undefined.")
        if condition_image_latents is None:
            raise ValueError("`condition_image_latents` is required: the reference's prior (pipe:647-656) needs the "
                             "condition image and crashes without it")
        if eta != 0.0:
            raise NotImplementedError("eta = 0 on the hot path (pipe:550)")
        if callback is not None and use_graph:
            raise ValueError("callbacks need use_graph=False (a replayed hipGraph has no per-step host hook)")
        dev = self.unet.device
        if dev.type != "cuda":
            raise HipLibraryError(f"unet is on {dev}: the HIP path has no CPU fallback")
        h_lat, w_lat = condition_image_latents.shape[-2:]
        height = height or h_lat * self.vae_scale_factor
        width = width or w_lat * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:                                                   # pipe:213-214
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
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

        cond_cpu = condition_image_latents.detach().to("cpu", torch.float32)
        latents = self.prepare_latents(batch_size, self.unet.config.in_channels, num_frames, height, width,
                                       torch.float32, dev, generator, latents)                  # pipe:635-645
        # first-frame-similarity prior (pipe:647-656), once per sample on the host with explicit generators
        blurred = gaussian_blur3(cond_cpu, blur_sigma)
        exp_blur = blurred.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        exp_cond = cond_cpu.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        mask = (torch.rand(exp_cond.shape, generator=prior_mask_generator)
                < frame_similarity_blurred_strength).float()
        prior = mask * exp_blur + (1 - mask) * exp_cond
        noise = torch.randn(prior.shape, generator=prior_noise_generator, dtype=torch.float32)
        latents = self.scheduler.add_noise(prior, noise, timesteps[0].repeat(batch_size))

        st = dict(
            latents=latents.to(dev).contiguous(), cond=cond_cpu.to(dev).contiguous(), copies=copies,
            num_frames=num_frames, guidance=float(guidance_scale),
            t_table=timesteps.to(torch.float32).to(dev), coef=self.scheduler.step_coefficients(timesteps).to(dev),
            step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
            ctx_text=prompt_embeds.to(dev, f16).contiguous(),
            ctx_ip=self.unet._project_image_embeds(
                {"image_embeds": image_embeds.to(dev)} if image_embeds is not None else None))
        if callback is None:
            self._run_steps(st, len(timesteps), use_graph)
        else:
            for i, t in enumerate(timesteps):                                                   # pipe:666-697
                self._step(st)
                if i % callback_steps == 0:
                    callback(i, t, st["latents"])
        latents = st["latents"]
        latents[:, 0] = st["cond"]                                                              # pipe:699-700
        if output_type != "latent":
            raise NotImplementedError("VAE decode is out of scope: use output_type='latent'")
        if not return_dict:
            return (latents,)
        return I2VAdapterPipelineOutput(frames=latents)

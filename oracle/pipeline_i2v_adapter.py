"""Oracle restatement of the denoising part of
/root/reference/src/pipelines/pipeline_i2v_adapter.py (TEST INFRASTRUCTURE).

Follows pipe:265-297 (prepare_latents), :529-536 (get_timesteps), :629-700 (prior, DDIM + CFG loop
with frame-0 re-injection).  The CLIP / VAE / PIL stages (pipe:300-527, 706-711) are out of scope
(SURVEY section 2 row 3b): the loop takes `prompt_embeds`, `negative_prompt_embeds`,
`condition_image_latents` (= vae.encode(img).sample() * scaling_factor, pipe:627) and optional
`image_embeds` directly.  RNG: the reference draws the prior mask / noise from the unseeded global RNG
(pipe:652,655); here they come from explicit generators.
"""
from typing import Optional

import torch

from .blocks import DDIMScheduler, gaussian_blur3


class I2VAdapterPipelineOutput:
    def __init__(self, frames):
        self.frames = frames


class I2VAdapterPipeline:
    vae_scale_factor = 8

    def __init__(self, unet, scheduler: Optional[DDIMScheduler] = None):
        self.unet = unet
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler()

    def get_timesteps(self, num_inference_steps, strength):
        """pipe:529-536."""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start:], num_inference_steps - t_start

    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, dtype, generator,
                        latents=None):
        """pipe:265-297."""
        shape = (batch_size, num_frames, num_channels_latents, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds, condition_image_latents, num_frames: int = 16,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, eta: float = 0.0, generator=None, latents=None,
                 image_embeds=None, negative_image_embeds=None, frame_similarity_sample_ratio: float = 1,
                 frame_similarity_blurred_strength: float = 0.6, prior_mask_generator=None,
                 prior_noise_generator=None, blur_sigma: Optional[float] = None, callback=None,
                 output_type="latent"):
        if condition_image_latents is None:
            raise ValueError("`condition_image_latents` is required (the reference crashes at pipe:648 without "
                             "a condition image)")
        h_lat, w_lat = condition_image_latents.shape[-2:]
        height = height or h_lat * self.vae_scale_factor
        width = width or w_lat * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        assert 0 < frame_similarity_sample_ratio <= 1
        batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])                    # pipe:613-614
            if image_embeds is not None:
                if negative_image_embeds is None:
                    negative_image_embeds = torch.zeros_like(image_embeds)                       # pipe:343
                image_embeds = torch.cat([negative_image_embeds, image_embeds])                  # pipe:621-622

        self.scheduler.set_timesteps(num_inference_steps)                                         # pipe:630-631
        timesteps, _ = self.get_timesteps(num_inference_steps, frame_similarity_sample_ratio)

        latents = self.prepare_latents(batch_size, self.unet.config.in_channels, num_frames, height, width,
                                       prompt_embeds.dtype, generator, latents)                   # pipe:635-645

        # first-frame-similarity prior, pipe:647-656.  torchvision GaussianBlur(kernel_size=3) (pipe:112) draws
        # sigma ~ U(0.1, 2.0) per call; here from the mask generator, before the mask
        if blur_sigma is None:
            blur_sigma = float(torch.empty(1).uniform_(0.1, 2.0, generator=prior_mask_generator).item())
        blurred = gaussian_blur3(condition_image_latents, blur_sigma)
        exp_blur = blurred.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        exp_cond = condition_image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        mask = (torch.rand(exp_cond.shape, generator=prior_mask_generator)
                < frame_similarity_blurred_strength).to(exp_cond.dtype)
        prior = mask * exp_blur + (1 - mask) * exp_cond
        noise = torch.randn(prior.shape, generator=prior_noise_generator, dtype=prior.dtype)
        latents = self.scheduler.add_noise(prior, noise, timesteps[0].repeat(batch_size))

        added = {"image_embeds": image_embeds} if image_embeds is not None else None
        for i, t in enumerate(timesteps):                                                         # pipe:666-697
            latents[:, 0] = condition_image_latents                                               # pipe:669
            x = torch.cat([latents] * 2) if do_cfg else latents
            x = self.scheduler.scale_model_input(x, t)
            noise_pred = self.unet(x, t, enable_cross_frame_attn=True, encoder_hidden_states=prompt_embeds,
                                   added_cond_kwargs=added).sample
            if do_cfg:
                u, c = noise_pred.chunk(2)
                noise_pred = u + guidance_scale * (c - u)                                         # pipe:686-688
            latents = self.scheduler.step(noise_pred, t, latents, eta=eta, generator=generator)   # pipe:659-660, 691
            if callback is not None:
                callback(i, t, latents)
        latents[:, 0] = condition_image_latents                                                   # pipe:699-700
        return I2VAdapterPipelineOutput(frames=latents)

"""Host-side image plumbing either side of the VAE (no arithmetic of the hot path): the subset of diffusers'
`VaeImageProcessor` the reference calls (`preprocess` at pipe:626, `postprocess` through `tensor2vid` pipe:53-65,711)
and `export_to_gif` (pipe:806-807)."""
from typing import List, Optional, Union

import numpy as np
import torch


class VaeImageProcessor:
    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True):
        self.vae_scale_factor, self.do_resize, self.do_normalize = vae_scale_factor, do_resize, do_normalize

    # ------------------------------------------------------------------ in
    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        """PIL image(s) / HWC uint8 or float numpy / NCHW tensor -> fp32 NCHW in [-1, 1], sized (height, width) rounded
        down to a multiple of the VAE scale factor."""
        import PIL.Image
        if isinstance(image, (PIL.Image.Image, np.ndarray, torch.Tensor)):
            image = [image]
        out = []
        for im in image:
            if isinstance(im, PIL.Image.Image):
                w0, h0 = im.size
                h = (height or h0) // self.vae_scale_factor * self.vae_scale_factor
                w = (width or w0) // self.vae_scale_factor * self.vae_scale_factor
                if self.do_resize and (w, h) != (w0, h0):
                    im = im.resize((w, h), resample=PIL.Image.LANCZOS)
                arr = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
                t = torch.from_numpy(arr).permute(2, 0, 1)
            elif isinstance(im, np.ndarray):
                arr = im.astype(np.float32) / (255.0 if im.dtype == np.uint8 else 1.0)
                t = torch.from_numpy(arr).permute(2, 0, 1) if arr.ndim == 3 else torch.from_numpy(arr)
            else:
                t = im.detach().float().cpu()
                if t.dim() == 4:
                    out.extend(self._normalize(x) for x in t)
                    continue
            out.append(self._normalize(t))
        return torch.stack(out)

    def _normalize(self, t):
        return 2.0 * t - 1.0 if self.do_normalize else t

    # ------------------------------------------------------------------ out
    @staticmethod
    def denormalize(images: torch.Tensor) -> torch.Tensor:
        return (images / 2 + 0.5).clamp(0, 1)

    def postprocess(self, image: torch.Tensor, output_type: str = "pil"):
        """NCHW tensor in [-1, 1] -> "pt" tensor in [0, 1] | "np" float NHWC | "pil" list of PIL images."""
        if output_type == "latent":
            return image
        image = self.denormalize(image.detach().float().cpu())
        if output_type == "pt":
            return image
        arr = image.permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            import PIL.Image
            return [PIL.Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
        raise ValueError(f"unsupported output_type {output_type}")


def tensor2vid(video: torch.Tensor, processor: VaeImageProcessor, output_type: str = "np") -> List:
    """pipe:53-65: (B, F, C, H, W) -> list over the batch of post-processed frame stacks."""
    return [processor.postprocess(video[b], output_type) for b in range(video.shape[0])]


def export_to_gif(frames, output_gif_path: str, fps: int = 8) -> str:
    """diffusers.utils.export_to_gif as called at pipe:807: a list of PIL frames -> animated GIF."""
    frames[0].save(output_gif_path, save_all=True, append_images=frames[1:], optimize=False,
                   duration=int(round(1000 / fps)), loop=0)
    return output_gif_path

"""The reference's evaluation driver (pipe:721-809) end to end on reduced models: checkpoint folders in the reference's
layout (config.json + safetensors, scheduler_config.json, ip-adapter_sd15.bin), a CSV of (image_path, name) pairs,
PNG condition images -> one GIF per prompt.  The CLIP encoders are out of scope: embeddings come from a file."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eval_driver_writes_gifs(dev, tmp_path):
    import PIL.Image
    from safetensors.torch import save_file
    import i2v_adapter_unofficial_amd as p
    from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
    from i2v_adapter_unofficial_amd.pipeline_i2v_adapter import main
    from tests.parity import sd15_ip_state_dict
    root = str(tmp_path)
    ch = (32, 64, 128, 128)
    kw = dict(sample_size=8, block_out_channels=ch, attention_head_dim=4, norm_num_groups=8, cross_attention_dim=64)
    u2 = init_random_weights_(p.UNet2DConditionModel(**kw), seed=1)
    u2.save_pretrained(os.path.join(root, "sd", "unet"))
    init_random_weights_(p.AutoencoderKL(block_out_channels=(32, 64, 64, 64), norm_num_groups=8), seed=2) \
        .save_pretrained(os.path.join(root, "sd", "vae"))
    # the SD-1.5 scheduler_config.json on disk says clip_sample / leading; the driver overrides both (pipe:755-757)
    p.DDIMScheduler().save_pretrained(os.path.join(root, "sd", "scheduler"))
    import json
    cfg_file = os.path.join(root, "sd", "scheduler", "scheduler_config.json")
    cfg = json.load(open(cfg_file))
    cfg.update(clip_sample=True, timestep_spacing="leading", _diffusers_version="0.24.0")
    json.dump(cfg, open(cfg_file, "w"))
    init_random_weights_(p.MotionAdapter(block_out_channels=ch, motion_num_attention_heads=4, motion_norm_num_groups=8),
                         seed=3).save_pretrained(os.path.join(root, "motion"))
    init_random_weights_(p.I2VAdapterModule(2, ch, 4), seed=4).save_pretrained(
        os.path.join(root, "checkpoint", "demo", "epoch_3", "i2v_adapter"))
    # IP-Adapter file: key ids follow the assembled model's attn_processors order
    probe = p.UNetMotionCrossFrameAttnModel.from_unet2d(u2, p.MotionAdapter(block_out_channels=ch, motion_num_attention_heads=4,
                                                                            motion_norm_num_groups=8), load_weights=False)
    os.makedirs(os.path.join(root, "ip", "models"))
    torch.save(sd15_ip_state_dict(probe, clip_dim=48), os.path.join(root, "ip", "models", "ip-adapter_sd15.bin"))
    os.makedirs(os.path.join(root, "data", "images"))
    rs = np.random.RandomState(0)
    names = ["a cat on a boat", "two dogs, running"]
    with open(os.path.join(root, "data", "eval.csv"), "w") as f:
        f.write("image_path,name\n")
        for i, nm in enumerate(names):
            PIL.Image.fromarray((rs.rand(70, 90, 3) * 255).astype("uint8")).save(os.path.join(root, "data", "images", f"{i}.png"))
            f.write(f'images/{i}.png,"{nm}"\n')
    g = torch.Generator().manual_seed(5)
    save_file({"prompt_embeds": torch.randn(2, 7, 64, generator=g), "negative_prompt_embeds": torch.randn(1, 7, 64, generator=g),
               "image_embeds": torch.randn(2, 48, generator=g)}, os.path.join(root, "embeds.safetensors"))
    rc = main(["--task_name", "demo", "--checkpoint_epoch", "3", "--eval_data_path", os.path.join(root, "data", "eval.csv"),
               "--embeds", os.path.join(root, "embeds.safetensors"), "--model_path", os.path.join(root, "sd"),
               "--motion_adapter_path", os.path.join(root, "motion"), "--ip_adapter_path", os.path.join(root, "ip"),
               "--checkpoint_root", os.path.join(root, "checkpoint"), "--samples_root", os.path.join(root, "samples"),
               "--num_frames", "4", "--num_inference_steps", "10"])
    assert rc == 0
    for nm in names:
        gif = PIL.Image.open(os.path.join(root, "samples", "demo", "epoch_3", f"{nm}.gif"))
        assert gif.n_frames == 4 and gif.size == (64, 64)
    assert main(["--embeds", "x"]) == -1           # the reference's "task_name must be specified" exit

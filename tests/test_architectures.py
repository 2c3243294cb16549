"""Known answers that pin the ARCHITECTURE restatements (layer set, names, shapes) to public facts — the arithmetic
of diffusers' UNet / VAE cannot be pinned here (diffusers is absent), but an architecture that is
rebuilt wrongly would not reproduce these totals:

  * runwayml/stable-diffusion-v1-5 UNet2DConditionModel: 859,520,964 parameters in 686 tensors
  * stabilityai/stable-diffusion-xl-base-1.0 UNet:       2,567,463,684 parameters in 1,680 tensors
  * AutoencoderKL decoder (49,490,179) + post_quant_conv (20) = 49,490,199
  * Salesforce/blip-image-captioning-large: state-dict names and shapes of transformers' own
    BlipForConditionalGeneration built from the same hyper-parameters (instantiated on the meta device)
"""
import pytest
import torch

from comat_amd import config, weights


def _count(sd):
    return sum(v.numel() for v in sd.values()), len(sd)


def test_unet_and_vae_parameter_counts():
    assert _count(weights.make_unet_weights(config.SD15_UNET, seed=None)) == (859_520_964, 686)
    assert _count(weights.make_unet_weights(config.SDXL_UNET, seed=None)) == (2_567_463_684, 1680)
    assert _count(weights.make_vae_weights(config.SD15_VAE, seed=None))[0] == 49_490_179 + 20
    r = config.SD15_UNET.lora_rank
    lora = weights.make_lora_weights(config.SD15_UNET)
    assert len(lora) == 32 * 4 * 2 and sum(v.numel() for v in lora.values()) == sum(
        r * (a + b) for (_, qd, kvd, inner) in weights.attention_names(config.SD15_UNET)
        for (a, b) in ((qd, inner), (kvd, inner), (kvd, inner), (inner, qd)))


def test_blip_large_state_dict_matches_transformers():
    tr = pytest.importorskip("transformers")
    c = config.BLIP_LARGE
    hf = tr.BlipConfig(
        vision_config=dict(hidden_size=c.v_hidden, intermediate_size=c.v_mlp, num_hidden_layers=c.v_layers,
                           num_attention_heads=c.v_heads, image_size=c.image_size, patch_size=c.patch_size),
        text_config=dict(hidden_size=c.t_hidden, intermediate_size=c.t_mlp, num_hidden_layers=c.t_layers,
                         num_attention_heads=c.t_heads, vocab_size=c.vocab_size, encoder_hidden_size=c.v_hidden,
                         max_position_embeddings=c.max_pos))
    with torch.device("meta"):
        model = tr.BlipForConditionalGeneration(hf)
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ours = {k: tuple(v.shape) for k, v in weights.make_blip_weights(c, seed=None).items()}
    # the LM head's decoder.{weight,bias} are aliases of the word embeddings / predictions.bias (tied), not extra tensors
    tied = {"text_decoder.cls.predictions.decoder.weight", "text_decoder.cls.predictions.decoder.bias"}
    assert set(ref) - tied == set(ours)
    assert all(ref[k] == ours[k] for k in ours)
    assert sum(p.numel() for p in model.parameters()) == sum(torch.Size(s).numel() for s in ours.values())

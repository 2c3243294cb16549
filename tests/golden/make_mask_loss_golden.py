"""Golden vectors for the attribute-concentration loss ASSEMBLY from the reference's OWN `GsamSegModel.get_mask_loss`
(+ `update_nouns_attributes`), run in the build container.

    python tests/golden/make_mask_loss_golden.py      # writes tests/golden/mask_loss.npz

`attr_concen_utils/gsam_interface.py` imports the two detector packages at the top (absent), so the two methods are pulled
out of the source with `ast` and executed as they are on a stand-in object; `get_grounding_loss_by_layer` is the reference's
own function (tc_loss_utils, imported through the torchvision shim of make_attn_golden.py).  The stand-ins: `get_mask(image,
nouns)` - the GroundingDINO + FastSAM call - hands back prepared masks (or None for a sample where the detector finds
nothing); `train_layer_ls`.  What the vectors pin beyond the per-layer loss (grounding_loss.npz): how the captured maps are
split per sample, which (timestep, layer) entries are read, which samples / objects are skipped, and the division by the batch
size."""
import ast
import os
import sys
import textwrap
import types
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import make_attn_golden as shimmed  # noqa: E402 - registers the torchvision shim, imports the reference's tc_loss_utils


def reference_methods(*names):
    src = open(os.path.join(REF, "attr_concen_utils", "gsam_interface.py")).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "defaultdict": defaultdict,
          "get_grounding_loss_by_layer": shimmed.ref_loss.get_grounding_loss_by_layer}
    for cls in (n for n in tree.body if isinstance(n, ast.ClassDef)):
        for fn in (n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names):
            exec(compile(textwrap.dedent(ast.get_source_segment(src, fn)), f"gsam_interface.py:{fn.name}", "exec"), ns)
    return [ns[n] for n in names]


def main():
    get_mask_loss, update_nouns_attributes = reference_methods("get_mask_loss", "update_nouns_attributes")
    g = torch.Generator().manual_seed(9)
    bs, heads, L, H = 3, 2, 16, 32
    layers = ["mid_4", "up_8"]
    attn_map = {}
    for ts in ("801", "401"):
        attn_map[ts] = {}
        for key, res, n in (("mid_4", 4, 1), ("up_8", 8, 3), ("down_8", 8, 2)):  # down_8: captured, never read
            attn_map[ts][key] = [torch.softmax(torch.randn(bs * heads, res, res, L, generator=g) * 2, dim=-1) for _ in range(n)]
    # sample 0: "a red car and a blue dog"; sample 1: the detector finds nothing; sample 2: "a cat, a cat and the sky over a
    # green skateboard" - the duplicated noun and the non-object noun are dropped, the split word stays
    subtrees = [[[2, 3], [6, 7]], [[2, 3]], [[1, 2], [4, 5], [8], [10, [11, 12]]]]
    pieces = [{1: "a", 2: "red", 3: "car", 4: "and", 5: "a", 6: "blue", 7: "dog"},
              {1: "a", 2: "tall", 3: "tree"},
              {1: "a", 2: "cat", 3: "a", 4: "big", 5: "cat", 6: "and", 7: "the", 8: "sky", 9: "a", 10: "green", 11: "skate", 12: "board"}]

    def box(y0, y1, x0, x1):
        m = torch.zeros(1, 1, H, H, dtype=torch.bool)
        m[..., y0:y1, x0:x1] = True
        return m
    mask_of = {"car": box(2, 14, 3, 20), "dog": box(16, 30, 10, 32), "skateboard": box(5, 28, 0, 12)}
    seen = []

    self = types.SimpleNamespace(train_layer_ls=layers)
    self.update_nouns_attributes = lambda nouns, attributes: update_nouns_attributes(self, nouns, attributes)

    def get_mask(image, nouns):
        seen.append(list(nouns))
        if "tree" in nouns:
            return None
        return [mask_of[n] for n in nouns]
    self.get_mask = get_mask
    images = torch.rand(bs, 3, H, H, generator=g)
    token_loss, pixel_loss, per_layer = get_mask_loss(self, images, None, subtrees, pieces, attn_map)
    out = {"bs": np.int64(bs), "heads": np.int64(heads), "layers": np.array(layers), "nouns_seen": np.array(["|".join(s) for s in seen]),
           "token_loss": np.float64(token_loss), "pixel_loss": np.float64(pixel_loss),
           "per_layer_keys": np.array(sorted(per_layer.keys()))}
    for ts in attn_map:
        for key, lst in attn_map[ts].items():
            out[f"map:{ts}:{key}"] = torch.stack(lst).numpy()
    for n, m in mask_of.items():
        out[f"mask:{n}"] = m.numpy()
    np.savez_compressed(os.path.join(HERE, "mask_loss.npz"), **out)
    print("token", float(token_loss), "pixel", float(pixel_loss), "detector prompts", seen, "entries", len(per_layer))


if __name__ == "__main__":
    main()

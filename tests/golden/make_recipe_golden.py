"""The reference's training recipes as its OWN argument parser reads them, run in the build container.

    python tests/golden/make_recipe_golden.py      # writes tests/golden/recipes.json

`training_utils/arguments.py` imports argparse only: `parse_args()` is called on the argument lists of scripts/sd15.sh and
scripts/sdxl.sh (everything after `training_script.py`), and the resulting namespaces - script values AND parser defaults -
are written out.  `comat_amd.step.StepConfig` (the path-relevant subset) is held against them in tests/test_step.py."""
import json
import os
import shlex
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from training_utils.arguments import parse_args  # noqa: E402


def script_argv(path):
    text = open(path).read().replace("\\\n", " ")
    line = next(l for l in text.splitlines() if "training_script.py" in l)
    toks = shlex.split(line)
    return toks[toks.index("training_script.py") + 1:]


def main():
    out = {}
    for name in ("sd15", "sdxl"):
        argv = script_argv(os.path.join(REF, "scripts", f"{name}.sh"))
        old = sys.argv
        sys.argv = ["training_script.py"] + argv
        try:
            ns = parse_args()
        finally:
            sys.argv = old
        out[name] = {k: v for k, v in sorted(vars(ns).items()) if isinstance(v, (int, float, str, bool, list, type(None)))}
    json.dump(out, open(os.path.join(HERE, "recipes.json"), "w"), indent=0, sort_keys=True)
    for name, d in out.items():
        print(name, {k: d[k] for k in ("learning_rate", "learning_rate_D", "max_grad_norm", "max_grad_norm_D", "K", "total_step",
                                       "resolution", "cfg_scale", "gan_loss", "gan_loss_weight", "adam_beta1", "adam_beta2",
                                       "adam_beta1_D", "adam_weight_decay", "adam_epsilon", "lora_rank", "mask_token_loss_weight",
                                       "mask_pixel_loss_weight", "attrcon_train_steps", "train_batch_size", "pretrain_model_name")
                     if k in d})


if __name__ == "__main__":
    main()

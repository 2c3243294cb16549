"""Golden vectors for the random choices of one optimisation step from the reference's OWN statements, run in the build
container.

    python tests/golden/make_step_sampling_golden.py      # writes tests/golden/step_sampling.json

`training_script.py` cannot be imported (accelerate / diffusers set-up at module level), and the choices are statements in
the middle of the training loop, not functions: the assignments that make them - the trained denoise steps
(training_script.py:563-566), the attribute-concentration steps (:589-590) and the crop of the decoded image (:606-609) - are
located in the file's syntax tree by their target names and executed as they are, in the file's order, on a seeded `random`
and an `args` namespace.  The same for the statements that compose the generator's loss from its terms (:618, :625, :639-640), on fixed scalar terms
and the weights of scripts/sd15.sh.  Nothing of them is stored in this repository, only their outputs: 20 seeds x 3
configurations of the choices, 4 compositions of the loss."""
import ast
import json
import os
import random
import types

REF = "/root/reference/training_script.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = ("interval", "max_start", "start", "training_steps", "offset_range", "random_offset_x", "random_offset_y", "size")


def statements():
    src = open(REF).read()
    found = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            seg = ast.get_source_segment(src, node)
            if isinstance(t, ast.Name) and t.id in WANTED and 556 <= node.lineno <= 612:
                found.append((node.lineno, seg))
            elif isinstance(t, ast.Subscript) and "attrcon_train_steps" in seg and "random.choices" in seg:
                found.append((node.lineno, seg))
    found.sort()
    assert [s.split("=")[0].strip() for _, s in found] == ["interval", "max_start", "start", "training_steps",
                                                            "kwargs['attrcon_train_steps']", "offset_range", "random_offset_x",
                                                            "random_offset_y", "size"], found
    return [compile(s, f"training_script.py:{ln}", "exec") for ln, s in found]


def loss_statements():
    """`loss = - caption_rewards["total"].mean()` and the three `loss += args.<weight> * <term>` that follow it"""
    src = open(REF).read()
    found = []
    for node in ast.walk(ast.parse(src)):
        if 612 <= getattr(node, "lineno", 0) <= 645:
            seg = ast.get_source_segment(src, node)
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id == "loss":
                found.append((node.lineno, "reward", seg))
            elif isinstance(node, ast.AugAssign) and isinstance(node.target, ast.Name) and node.target.id == "loss":
                kind = "gan" if "G_loss" in seg else ("token" if "token_loss" in seg else "pixel")
                found.append((node.lineno, kind, seg))
    found.sort()
    assert [k for _, k, _ in found] == ["reward", "gan", "token", "pixel"], found
    return {k: compile(s, f"training_script.py:{ln}", "exec") for ln, k, s in found}


def main():
    import torch
    loss_codes = loss_statements()
    losses = []
    for gan, attrcon in ((False, False), (True, False), (False, True), (True, True)):
        terms = dict(reward=[-2.75, -3.5], G_loss=0.8125, token_loss=1.375, pixel_loss=6.5)
        ns = {"args": types.SimpleNamespace(gan_loss_weight=1.0, mask_token_loss_weight=1e-3, mask_pixel_loss_weight=5e-5),
              "caption_rewards": {"total": torch.tensor(terms["reward"])}, "G_loss": torch.tensor(terms["G_loss"]),
              "token_loss": torch.tensor(terms["token_loss"]), "pixel_loss": torch.tensor(terms["pixel_loss"])}
        for kind in ["reward"] + (["gan"] if gan else []) + (["token", "pixel"] if attrcon else []):
            exec(loss_codes[kind], ns)
        losses.append(dict(gan=gan, attrcon=attrcon, terms=terms, gan_loss_weight=1.0, mask_token_loss_weight=1e-3,
                           mask_pixel_loss_weight=5e-5, loss=float(ns["loss"])))
    codes = statements()
    out = []
    for total_step, K, res, n_attr in ((50, 5, 512, 2), (5, 5, 512, 2), (50, 5, 1024, 2)):
        for seed in range(20):
            random.seed(seed)
            ns = {"random": random, "total_step": total_step, "kwargs": {},
                  "args": types.SimpleNamespace(K=K, resolution=res, attrcon_train_steps=n_attr)}
            for c in codes:
                exec(c, ns)
            out.append(dict(total_step=total_step, K=K, resolution=res, attrcon_train_steps=n_attr, seed=seed,
                            training_steps=ns["training_steps"], attrcon_steps=ns["kwargs"]["attrcon_train_steps"],
                            offset_x=ns["random_offset_x"], offset_y=ns["random_offset_y"], size=ns["size"]))
    json.dump(dict(sampling=out, loss=losses), open(os.path.join(HERE, "step_sampling.json"), "w"), indent=0)
    print(len(out), "cases; first:", out[0])
    print("loss compositions:", [(c["gan"], c["attrcon"], c["loss"]) for c in losses])


if __name__ == "__main__":
    main()

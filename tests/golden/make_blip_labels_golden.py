"""Golden vectors for the caption-loss LABELS from the reference's OWN `Blip.score`, run in the build container.

    python tests/golden/make_blip_labels_golden.py      # writes tests/golden/blip_labels.json

`concept_mat_utils/caption_blip.py` imports torchvision at the top (absent), so the method is pulled out of the source with `ast`
and executed as it is on a stand-in object: the image transform is the identity, the processor hands back prepared token ids,
and the model call only records what it is given and returns a fixed loss.  What the vectors pin: the text handed to the
tokenizer ('a photography of ' + the lower-cased prompt), which positions of the ids become labels (pads and the first
`prompt_length` positions are ignored, caption_blip.py:51-54) and reward = -loss.  (`prompt_length` is computed in `__init__`
from the BERT tokenizer, absent offline: '[CLS] a photography of [SEP]' is 5 ids, minus 1 = 4.)"""
import ast
import json
import os
import textwrap
import types

import torch

REF = "/root/reference/concept_mat_utils/caption_blip.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "Blip")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "score")
    ns = {"torch": torch}
    exec(compile(textwrap.dedent(ast.get_source_segment(src, fn)), "caption_blip.py:score", "exec"), ns)
    ids = torch.tensor([[101, 1037, 5855, 1997, 2417, 2482, 102, 0, 0],
                        [101, 1037, 5855, 1997, 1037, 2630, 3899, 2006, 102],
                        [101, 1037, 5855, 1997, 4937, 102, 0, 0, 0]])
    seen = {}

    def processor(images=None, text=None, return_tensors=None, padding=None):
        seen["text"], seen["padding"] = list(text), padding
        return {"pixel_values": images, "input_ids": ids.clone(), "attention_mask": (ids != 0).long()}
    processor.tokenizer = types.SimpleNamespace(pad_token_id=0)

    def model(**inputs):
        seen["labels"] = inputs["labels"].clone()
        seen["keys"] = sorted(inputs)
        return types.SimpleNamespace(loss=torch.tensor(1.625))
    self = types.SimpleNamespace(transforms=lambda im: im, prompt="a photography of", prompt_length=4, processor=processor, model=model)
    images = torch.rand(3, 3, 4, 4)
    reward = ns["score"](self, images, ["A Red Car", "a blue dog on", "Cat"])
    out = dict(input_ids=ids.tolist(), pad_token_id=0, prompt_length=4, labels=seen["labels"].tolist(), text=seen["text"],
               model_inputs=seen["keys"], loss=1.625, reward=float(reward))
    json.dump(out, open(os.path.join(HERE, "blip_labels.json"), "w"), indent=0)
    print(out)


if __name__ == "__main__":
    main()

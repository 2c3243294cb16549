"""comat_amd/attr_index.py (index half) against tests/golden/attr_index.json = the outputs of
the reference's own functions on the same hand-written parses (tests/golden/make_attr_index_golden.py: the three
extractors of attribute_concen_utils.py, unify_lists + _align_indices of AttrConcenTrainableSDPipeline.py,
get_attention_map_index_to_wordpiece, update_nouns_attributes of gsam_interface.py) - including the reference's quirks
(the verbs extractor examines one noun only; an auxiliary-rooted group ends in its modifier, which then names the object)."""
import json
import os

import pytest

from comat_amd import attr_index as A

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attr_index.json")))


class Tok:
    def __init__(self, text, pos, dep):
        self.text, self.pos_, self.dep_, self.children = text, pos, dep, []


def build(case):
    toks = [Tok(t, p, d) for t, p, d, _ in case["spec"]]
    for i, (_, _, _, h) in enumerate(case["spec"]):
        if h >= 0:
            toks[h].children.append(toks[i])
    pcs = [A.START_TOKEN]
    for w in case["words"]:
        parts = w.split("|")
        pcs += parts[:-1] + [parts[-1] + "</w>"]
    pcs.append(A.END_TOKEN)
    return toks, dict(enumerate(pcs))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_matches_reference_functions(name):
    case = GOLD[name]
    doc, pieces = build(case)
    idx = {id(t): i for i, t in enumerate(doc)}
    ids = lambda groups: [[idx[id(t)] for t in g] for g in groups]
    assert ids(A.noun_modifier_groups(doc)) == case["g1"]
    assert ids(A.groups_below_auxiliaries(doc)) == case["g2"]
    assert ids(A.noun_modifier_groups_through_verbs(doc)) == case["g3"]
    groups = A.extract_groups(doc)
    assert ids(groups) == case["groups"]
    aligned = A.align_groups(pieces, groups)
    assert aligned == case["aligned"]
    assert {str(k): v for k, v in A.position_to_piece(pieces).items()} == case["pos2piece"]
    nouns, attrs = A.attribute_token_lists(doc, pieces)
    assert nouns == case["nouns"] and attrs == case["attributes"]


def test_attribute_lists_feed_the_grounding_loss(sim):
    """the `attributes` this module returns are what losses.mask_loss takes (token positions < 77, one list per mask)"""
    import numpy as np
    import torch

    from comat_amd import losses
    doc, pieces = build(GOLD["two_objects"])
    nouns, attrs = A.attribute_token_lists(doc, pieces)
    assert nouns == ["car", "dog"] and all(0 < t < 77 for a in attrs for t in a)
    g = torch.Generator().manual_seed(0)
    amap = torch.softmax(torch.randn(2, 4, 4, 77, generator=g), -1)  # (heads, res, res, L) of one sample, one layer
    masks = np.zeros((2, 16, 16), dtype=bool)
    masks[0, :8], masks[1, 8:] = True, True
    tl, pl = losses.mask_loss({"801": {"up_4": [amap]}}, [masks], [attrs], ("up_4",), 1, sim)
    assert torch.isfinite(tl) and torch.isfinite(pl) and float(tl) > 0 and float(pl) > 0

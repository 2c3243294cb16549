"""Golden vectors for comat_amd/attr_index.py from the reference's OWN functions, run in the build container.

    python tests/golden/make_attr_index_golden.py      # writes tests/golden/attr_index.json

`attribute_concen_utils.py` imports cleanly (torch only).  `_align_indices`, `unify_lists` / `is_sublist`
(AttrConcenTrainableSDPipeline.py) and the noun / attribute assembly + `update_nouns_attributes`
(attr_concen_utils/gsam_interface.py) live in modules that import diffusers / ultralytics at the top, which are absent:
their function definitions are pulled out of the source with `ast` and executed here as they are - nothing of them is
stored in this repository, only their outputs.  Inputs are hand-written dependency parses (what spaCy's
en_core_web_trf would hand over: the parser network is absent too) and word-piece tables in CLIP's format."""
import ast
import json
import os
import sys
import textwrap

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


class Tok:
    def __init__(self, text, pos, dep):
        self.text, self.pos_, self.dep_, self.children = text, pos, dep, []

    def __repr__(self):
        return self.text


def parse(spec):
    """spec: [(text, pos, dep, head index or -1)] -> list of Tok with .children in sentence order"""
    toks = [Tok(t, p, d) for t, p, d, _ in spec]
    for i, (_, _, _, h) in enumerate(spec):
        if h >= 0:
            toks[h].children.append(toks[i])
    return toks


def pieces(words):
    """CLIP-style pieces: a list entry 'skate|board' is one word split in two pieces"""
    out = ["<|startoftext|>"]
    for w in words:
        parts = w.split("|")
        out += parts[:-1] + [parts[-1] + "</w>"]
    return out + ["<|endoftext|>"]


CASES = {
    # a red car and a blue dog
    "two_objects": ([("a", "DET", "det", 2), ("red", "ADJ", "amod", 2), ("car", "NOUN", "ROOT", -1), ("and", "CCONJ", "cc", 2),
                     ("a", "DET", "det", 6), ("blue", "ADJ", "amod", 6), ("dog", "NOUN", "conj", 2)],
                    ["a", "red", "car", "and", "a", "blue", "dog"]),
    # a wooden skateboard next to a fluffy white cat  (split word, two modifiers)
    "split_word": ([("a", "DET", "det", 2), ("wooden", "ADJ", "amod", 2), ("skateboard", "NOUN", "ROOT", -1),
                    ("next", "ADV", "advmod", 2), ("to", "ADP", "prep", 3), ("a", "DET", "det", 8), ("fluffy", "ADJ", "amod", 8),
                    ("white", "ADJ", "amod", 8), ("cat", "NOUN", "pobj", 4)],
                   ["a", "wooden", "skate|board", "next", "to", "a", "fluffy", "white", "cat"]),
    # the car is red and the sky is blue  (auxiliary roots; 'sky' is not an object)
    "aux_root": ([("the", "DET", "det", 1), ("car", "NOUN", "nsubj", 2), ("is", "AUX", "ROOT", -1), ("red", "ADJ", "acomp", 2),
                  ("and", "CCONJ", "cc", 2), ("the", "DET", "det", 6), ("sky", "NOUN", "nsubj", 7), ("is", "AUX", "conj", 2),
                  ("blue", "ADJ", "acomp", 7)],
                 ["the", "car", "is", "red", "and", "the", "sky", "is", "blue"]),
    # a dog that is brown  (relative clause through a verb: only reachable through the first-token quirk)
    "relcl": ([("dog", "NOUN", "ROOT", -1), ("that", "PRON", "nsubj", 2), ("is", "AUX", "relcl", 0), ("brown", "ADJ", "acomp", 2)],
              ["dog", "that", "is", "brown"]),
    # a red apple and a green apple  (the same noun twice: both dropped by the duplicate rule)
    "duplicate_noun": ([("a", "DET", "det", 2), ("red", "ADJ", "amod", 2), ("apple", "NOUN", "ROOT", -1), ("and", "CCONJ", "cc", 2),
                        ("a", "DET", "det", 6), ("green", "ADJ", "amod", 6), ("apple", "NOUN", "conj", 2)],
                       ["a", "red", "apple", "and", "a", "green", "apple"]),
    # a big old red brick house (four modifiers: the group is too long and disappears), a small bird
    "too_long": ([("a", "DET", "det", 5), ("big", "ADJ", "amod", 5), ("old", "ADJ", "amod", 5), ("red", "ADJ", "amod", 5),
                  ("brick", "NOUN", "compound", 5), ("house", "NOUN", "ROOT", -1), ("a", "DET", "det", 8),
                  ("small", "ADJ", "amod", 8), ("bird", "NOUN", "appos", 5)],
                 ["a", "big", "old", "red", "brick", "house", "a", "small", "bird"]),
    # red red bear and a very shiny motorcycle (repeated modifier; adverb below the adjective; three-piece word)
    "repeats": ([("red", "ADJ", "amod", 2), ("red", "ADJ", "amod", 2), ("bear", "NOUN", "ROOT", -1), ("and", "CCONJ", "cc", 2),
                 ("a", "DET", "det", 7), ("very", "ADV", "advmod", 6), ("shiny", "ADJ", "amod", 7), ("motorcycle", "NOUN", "conj", 2)],
                ["red", "red", "bear", "and", "a", "very", "shiny", "mo|tor|cycle"]),
}


def functions_of(path, names, methods=()):
    """source text of the named module-level functions (and of methods, dedented and stripped of `self`) -> namespace"""
    src = open(path).read()
    tree = ast.parse(src)
    ns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names + tuple(methods):
            code = textwrap.dedent(ast.get_source_segment(src, node))
            exec(compile(code, path, "exec"), ns)  # noqa: S102 - the reference's own code, executed to produce fixtures
    return ns


def main():
    import attribute_concen_utils as R
    ns = functions_of(os.path.join(REF, "AttrConcenTrainableSDPipeline.py"), ("is_sublist", "unify_lists"), ("_align_indices",))
    ns.update(start_token=R.start_token, end_token=R.end_token, align_wordpieces_indices=R.align_wordpieces_indices)
    gs = functions_of(os.path.join(REF, "attr_concen_utils", "gsam_interface.py")
                      if os.path.exists(os.path.join(REF, "attr_concen_utils", "gsam_interface.py"))
                      else next(os.path.join(d, "gsam_interface.py") for d, _, fs in os.walk(REF) if "gsam_interface.py" in fs),
                      (), ("update_nouns_attributes",))
    from collections import defaultdict
    gs["defaultdict"] = defaultdict
    out = {}
    for name, (spec, words) in CASES.items():
        doc = parse(spec)
        pcs = pieces(words)
        table = {i: p for i, p in enumerate(pcs)}

        class Tokz:  # what get_indices needs of a tokenizer
            def __call__(self, prompt):
                return type("E", (), {"input_ids": list(range(len(pcs)))})()

            def convert_ids_to_tokens(self, ids):
                return [pcs[i] for i in ids]
        ns["get_indices"] = lambda tokenizer, prompt: R.get_indices(tokenizer, prompt)
        holder = type("P", (), {"tokenizer": Tokz()})()
        g1 = R.extract_attribution_indices(doc) or []
        g2 = R.extract_attribution_indices_with_verb_root(doc) or []
        g3 = R.extract_attribution_indices_with_verbs(doc) or []
        merged = [p for p in ns["unify_lists"](g1, g2, g3) if len(p) < 4]
        aligned = ns["_align_indices"](holder, "prompt", merged)
        pos2piece = R.get_attention_map_index_to_wordpiece(Tokz(), "prompt")
        # the noun / attribute assembly of get_mask_loss (gsam_interface.py:167-190) is inline code, not a function: its
        # few lines are restated here, the filter that follows is the reference's own update_nouns_attributes
        nouns, attrs = [], []
        for sub in aligned:
            if len(sub) < 1:
                continue
            npos = sub[-1] if isinstance(sub[-1], list) else [sub[-1]]
            nouns.append("".join(pos2piece[i] for i in npos))
            flat = []
            for it in sub[:-1]:
                flat.extend(it if isinstance(it, list) else [it])
            attrs.append(flat + list(npos))
        fn, fa = gs["update_nouns_attributes"](None, nouns, attrs) if nouns else ([], [])
        idx = {id(t): i for i, t in enumerate(doc)}
        out[name] = dict(spec=spec, words=words, groups=[[idx[id(t)] for t in g] for g in merged],
                         g1=[[idx[id(t)] for t in g] for g in g1], g2=[[idx[id(t)] for t in g] for g in g2],
                         g3=[[idx[id(t)] for t in g] for g in g3], aligned=aligned,
                         pos2piece={str(k): v for k, v in pos2piece.items()}, nouns=fn, attributes=fa)
    with open(os.path.join(HERE, "attr_index.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, v["groups"], v["aligned"], v["nouns"], v["attributes"])


if __name__ == "__main__":
    main()

"""Attribute / noun token lists for the attribute-concentration loss (index half).

The grounding loss (`comat_amd/losses.py`) takes, per prompt, `attributes` = one list of text-token indices per object
("a red car" -> the positions of `red` and `car` in the 77-token CLIP sequence) and one mask per object.  In the
reference those come from two providers that sit in front of the hot path:

  * indices: a spaCy dependency parse -> (modifiers..., noun) groups (`attribute_concen_utils.py:39-131`), the three
    extractors merged and pruned (`AttrConcenTrainableSDPipeline.py:281-296,539-564`), aligned with the tokenizer's
    word pieces (`:298-338`, `attribute_concen_utils.py:10-36`), then turned into noun strings + flat index lists with
    duplicate / non-object nouns dropped (`attr_concen_utils/gsam_interface.py:167-196,232-262`);
  * masks: GroundingDINO + FastSAM on the decoded image, prompted with those noun strings (`gsam_interface.py:54-137`).

This module is the first provider's arithmetic on plain data: it needs a parse (any sequence of tokens with `.text`,
`.pos_`, `.dep_`, `.children` - spaCy's `Doc` qualifies; the parser model `en_core_web_trf` itself is a third-party
network, absent offline) and the tokenizer's word pieces (`{position: piece}`; CLIP pieces end a word with `</w>`).  The
mask provider stays out of scope: its two detector networks are inputs to the path, and its
interface here is the noun list this module returns.  Pure host code, no tensors.

Pinned by `tests/golden/attr_index.json` = outputs of the reference's own functions (run in the build container by
`tests/golden/make_attr_index_golden.py`) on hand-written parses.
"""
from __future__ import annotations

START_TOKEN, END_TOKEN = "<|startoftext|>", "<|endoftext|>"
_MODIFIERS = ("amod", "nmod", "compound", "npadvmod", "advmod", "acomp")
_NOUNS = ("NOUN", "PROPN")
# nouns that name no groundable object: the reference drops them (and their plural minus one letter) before it asks the
# detector for masks (gsam_interface.py:247-252)
NON_OBJECT_NOUNS = frozenset((
    "scene surface area atmosphere noise place kitchen dream interior exterior meal background bathroom room scent street "
    "hillside mountain sky sea ocean lost language skill one night day morning space environment conditions field shore "
    "restroom party grass snow meadow water shadow waves song cycle sunlight mysteries wall salon range cry speech tone "
    "thing about activity air advertisement airport also").split())


# ---- (modifiers..., noun) groups from a dependency parse ------------------------------------------------------------------
def _descend(stack, keep, group, modifiers):
    """depth-first over `stack` (LIFO, children appended in order): a node that is a modifier or a conjunct continues the
    walk through its children and joins the group when `keep(node)`"""
    while stack:
        node = stack.pop()
        if node.dep_ in modifiers or node.dep_ == "conj":
            if keep(node):
                group.append(node)
            stack.extend(node.children)


def noun_modifier_groups(doc):
    """every noun that is not itself a modifier, with the modifiers hanging below it: [modifier, ..., noun]
    (`extract_attribution_indices`)"""
    groups = []
    for w in doc:
        if w.pos_ not in _NOUNS or w.dep_ in _MODIFIERS:
            continue
        group, stack = [], []
        for child in w.children:
            if child.dep_ in _MODIFIERS:
                group.append(child)
                stack.extend(child.children)
        _descend(stack, lambda n: True, group, _MODIFIERS)
        if group:
            groups.append(group + [w])
    return groups


def noun_modifier_groups_through_verbs(doc):
    """"a dog that is red": relative clauses count as modifiers, verbs / auxiliaries are walked through but not kept
    (`extract_attribution_indices_with_verbs`).  The reference returns from INSIDE its loop over the tokens, so only the
    first token of the parse is ever examined - reproduced, because the golden vectors and any
    checkpoint trained against the reference see exactly that."""
    modifiers = _MODIFIERS + ("relcl",)
    not_verb = lambda n: n.pos_ not in ("AUX", "VERB")
    for w in doc:
        if w.pos_ not in _NOUNS or w.dep_ in modifiers:
            continue
        group, stack = [], []
        for child in w.children:
            if child.dep_ in modifiers:
                if not_verb(child):
                    group.append(child)
                stack.extend(child.children)
        _descend(stack, not_verb, group, modifiers)
        return [group + [w]] if group else []
    return []


def groups_below_auxiliaries(doc):
    """"the car is red": an auxiliary whose children hold a noun and a modifier (`extract_attribution_indices_with_verb_root`)"""
    groups = []
    for w in doc:
        if w.pos_ != "AUX" or w.dep_ in _MODIFIERS:
            continue
        group, stack = [], []
        for child in w.children:
            if child.dep_ in _MODIFIERS or child.pos_ in _NOUNS:
                if child.pos_ not in ("AUX", "VERB"):
                    group.append(child)
                stack.extend(child.children)
        if len(group) < 2:
            continue
        _descend(stack, lambda n: n.pos_ != "AUX", group, _MODIFIERS)
        groups.append(group)  # (the auxiliary itself is never part of the group)
    return groups


def merge_groups(*group_lists, max_len=3):
    """one list without duplicates and without groups wholly contained in a longer one, shortest first
    (`unify_lists`), then only groups of at most `max_len` members (`_extract_attribution_indices`: len < 4)"""
    every = sorted((g for gl in group_lists for g in gl), key=len)
    out, seen = [], set()
    for i, g in enumerate(every):
        key = tuple(id(t) for t in g)
        if key in seen:
            continue
        inside_longer = any(len(g) < len(h) and all(any(t is u for u in h) for t in g) for h in every[i + 1:])
        if not inside_longer:
            out.append(g)
            seen.add(key)
    return [g for g in out if len(g) <= max_len]


def extract_groups(doc):
    return merge_groups(noun_modifier_groups(doc), groups_below_auxiliaries(doc), noun_modifier_groups_through_verbs(doc))


# ---- alignment with the tokenizer's word pieces ------------------------------------------------------------------------------
def _bare(piece):
    return piece.replace("</w>", "")


def continue_word(pieces, start, word):
    """positions start, start+1, ... whose pieces spell `word`; [] if they do not (`align_wordpieces_indices`)"""
    got, spelled = [start], _bare(pieces[start])
    for pos in range(start + 1, len(pieces)):
        if spelled == word:
            break
        nxt = _bare(pieces[pos])
        if word.startswith(spelled + nxt) and nxt != word:
            spelled += nxt
            got.append(pos)
        else:
            return []
    return got


def align_groups(pieces, groups):
    """pieces: {position: word piece} of the prompt (positions 0 .. n-1, first = start token, last = end token).
    -> per group the positions of its members, a bare int for a one-piece word and a list for a split word, each
    position used at most once over all groups (`_align_indices`)"""
    used, out = set(), []
    for group in groups:
        mine = []
        for member in group:
            for pos, piece in pieces.items():
                if piece in (START_TOKEN, END_TOKEN):
                    continue
                bare = _bare(piece)
                if member.text == bare:
                    if pos not in mine and pos not in used:
                        mine.append(pos)
                        break
                elif member.text.startswith(bare) and bare != member.text:
                    span = continue_word(pieces, pos, member.text)
                    if span and span not in mine and all(p not in used for p in span):
                        mine.append(span)
                        break
        for item in mine:
            used.update(item if isinstance(item, list) else [item])
        out.append(mine)
    return out


def position_to_piece(pieces):
    """{position: bare piece} without the start / end tokens (`get_attention_map_index_to_wordpiece`)"""
    keys = list(pieces)
    return {k: _bare(pieces[k]) for k in keys[1:-1]}


# ---- what the loss and the mask provider consume --------------------------------------------------------------------------------
def nouns_and_attributes(aligned_groups, pos2piece):
    """-> (nouns, attributes): per object its noun string (the detector prompt) and the flat list of token positions of
    its modifiers and of the noun itself (`get_mask_loss`, gsam_interface.py:167-190), then duplicates of a noun are
    dropped altogether and so are nouns that name no object (`update_nouns_attributes`)"""
    nouns, attrs = [], []
    for group in aligned_groups:
        if len(group) < 1:
            continue
        noun_pos = group[-1] if isinstance(group[-1], list) else [group[-1]]
        nouns.append("".join(pos2piece[p] for p in noun_pos))
        flat = []
        for item in group[:-1]:
            flat.extend(item if isinstance(item, list) else [item])
        attrs.append(flat + list(noun_pos))
    where = {}
    for i, n in enumerate(nouns):
        where.setdefault(n, []).append(i)
    keep = [(n, attrs[ix[0]]) for n, ix in where.items() if len(ix) == 1]
    keep = [(n, a) for n, a in keep if n not in NON_OBJECT_NOUNS and n[:-1] not in NON_OBJECT_NOUNS]
    return [n for n, _ in keep], [a for _, a in keep]


def attribute_token_lists(doc, pieces):
    """parse + word pieces of ONE prompt -> (nouns for the mask provider, `attributes` for `losses.mask_loss`)"""
    return nouns_and_attributes(align_groups(pieces, extract_groups(doc)), position_to_piece(pieces))

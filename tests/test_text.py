"""CPU tests: tokeniser parity with the reference's word_utils.Corpus (golden made by
oracle/gen_golden.py from /root/reference/dmn_pytorch/utils/word_utils.py)."""
import json
import os

import pytest
import torch

from dms_amd.text import Corpus, PAD_TOKEN, UNK_TOKEN

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenize.json")))


def _corpus():
    c = Corpus()
    for l in G["vocab_lines"]:
        c.add_to_corpus(l)
    c.dictionary.add_word(UNK_TOKEN)
    c.dictionary.add_word(PAD_TOKEN)
    return c


def test_vocabulary_matches_reference():
    c = _corpus()
    assert c.dictionary.idx2word == G["idx2word"] and len(c) == len(G["idx2word"])
    assert c.dictionary["horse"] == G["idx2word"].index("horse") and "horse" in c.dictionary
    assert c.dictionary[0] == G["idx2word"][0] and c.dictionary[[0, 1]] == G["idx2word"][:2]


@pytest.mark.parametrize("max_len,key", [(20, "ids_20"), (5, "ids_5"), (-1, "ids_nolimit")])
def test_tokenize_matches_reference(max_len, key):
    c = _corpus()
    for q, want in zip(G["queries"], G[key]):
        got = c.tokenize(q, max_len)
        assert got.dtype == torch.long and got.tolist() == want, q


def test_quirks_left_pad_separators_and_batch():
    c = _corpus()
    d = c.dictionary
    ids = c.tokenize("the left horse", 8).tolist()
    # separators are tokens (-> <unk>), pads go on the LEFT
    assert ids == [d[PAD_TOKEN]] * 3 + [d["the"], d[UNK_TOKEN], d["left"], d[UNK_TOKEN], d["horse"]]
    assert c.tokenize("horse .", -1).tolist() == [d["horse"], d[UNK_TOKEN]]   # '.' dropped, ' ' kept
    b = c.tokenize_batch(G["queries"], 20)
    assert b.shape == (len(G["queries"]), 20) and b.tolist() == G["ids_20"]
    with pytest.raises(IndexError):
        c.tokenize("   ")

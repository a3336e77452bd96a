"""CPU tests: readers of the reference's on-disk query formats (dms_amd/data.py) against files
written by the reference's OWN classes (tests/golden/refdata, made by `oracle/gen_golden.py
datafiles`: word_utils.Corpus torch.save'd under its real module path, split lists of triples)."""
import json
import os
import pickle

import pytest
import torch

from dms_amd import data
from dms_amd.text import Corpus, Dictionary, PAD_TOKEN

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata")
EXP = json.load(open(os.path.join(ROOT, "expected.json")))


@pytest.mark.parametrize("fname", ["corpus.pth", "corpus_legacy.pth"])
def test_corpus_file_is_read_without_the_reference_package(fname):
    c = data.load_corpus(os.path.join(ROOT, "unc", fname))
    assert type(c) is Corpus and type(c.dictionary) is Dictionary
    assert c.dictionary.idx2word == EXP["unc_idx2word"] and c.max_len == 20
    assert c.dictionary["horse"] == EXP["unc_idx2word"].index("horse")
    for rec in (EXP["unc_train"], EXP["unc_val"]):
        for (_, _, phrase), want20, want8 in zip(rec["triples"], rec["ids_20"], rec["ids_8"]):
            assert c.tokenize(phrase, 20).tolist() == want20
            assert c.tokenize(phrase, 8).tolist() == want8


def test_split_lists_and_query_set_mirror_refer_dataset():
    val = data.load_split(os.path.join(ROOT, "unc", "unc_val.pth"))
    assert val == [tuple(t) for t in EXP["unc_val"]["triples"]]
    qs = data.QuerySet(ROOT, "unc", "trainval")                       # = train + val (referit_loader.py:91-97)
    assert len(qs) == len(EXP["unc_train"]["triples"]) + len(val)
    assert qs.images[:5] == [tuple(t) for t in EXP["unc_train"]["triples"]]
    ids = qs.tokenize_phrase(qs.phrase(0))
    assert ids.tolist() == EXP["unc_train"]["ids_20"][0]
    assert qs.untokenize_word_vector(ids.tolist())[-1] == "horse"
    b = qs.query_batch(range(len(qs)))
    assert b.shape == (len(qs), 20) and b.dtype == torch.long
    assert b.tolist() == EXP["unc_train"]["ids_20"] + EXP["unc_val"]["ids_20"]
    assert b[3, 0].item() == qs.corpus.dictionary[PAD_TOKEN]            # pads on the left
    ref = data.QuerySet(ROOT, "referit", "val", max_query_len=8)
    assert ref.query_batch([0, 4]).tolist() == [EXP["referit_val"]["ids_8"][0], EXP["referit_val"]["ids_8"][4]]
    with pytest.raises(ValueError, match="does not have split"):
        data.QuerySet(ROOT, "gref", "testA")
    with pytest.raises(ValueError, match="not supported"):
        data.QuerySet(ROOT, "coco", "val")


class _Evil(object):
    def __reduce__(self):
        return (os.system, ("echo pwned > /tmp/dmn_pwned",))


def test_files_that_hold_code_are_refused(tmp_path):
    """The readers never unpickle arbitrary objects: a file whose pickle names any global besides
    the two vocabulary classes is rejected and nothing in it runs."""
    p = str(tmp_path / "evil.pth")
    torch.save([("a.jpg", "1.pth", "phrase"), _Evil()], p)
    if os.path.exists("/tmp/dmn_pwned"):
        os.remove("/tmp/dmn_pwned")
    with pytest.raises(pickle.UnpicklingError):
        data.load_split(p)
    with pytest.raises(pickle.UnpicklingError):
        data.load_corpus(p)
    assert not os.path.exists("/tmp/dmn_pwned")
    q = str(tmp_path / "wrong.pth")
    torch.save([("a.jpg", "1.pth")], q)
    with pytest.raises(ValueError, match="string triple"):
        data.load_split(q)
    with pytest.raises(ValueError, match="does not hold a word_utils.Corpus"):
        data.load_corpus(q)

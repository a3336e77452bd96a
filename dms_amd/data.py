# -*- coding: utf-8 -*-
"""Readers for the reference's on-disk query formats (host side; feeds ``lang``).

``ReferDataset`` (dmn_pytorch/referit_loader.py:36-233) keeps, per dataset, under
``<split_root>/<dataset>/``:

* ``corpus.pth``              -- ``torch.save`` of a ``word_utils.Corpus`` object (its
  ``dictionary.word2idx / idx2word`` and ``max_len``; referit_loader.py:134-138, 172-176),
* ``<dataset>_<split>.pth``   -- ``torch.save`` of a list of ``(image file, mask file, phrase)``
  string triples (referit_loader.py:153-157, 194-199), loaded at :89-97.

Both are pickles.  They are read here with ``torch.load(..., weights_only=True)`` ONLY: the
restricted unpickler builds containers, strings and numbers and refuses every global except
the two vocabulary classes, which are mapped onto ``dms_amd.text.Corpus / Dictionary`` by
their pickled names -- nothing from the file is ever executed, and the reference package is
not needed to read its files.  A file holding anything else raises.

``QuerySet`` mirrors the text half of ``ReferDataset`` (same attribute and method names:
``corpus``, ``images``, ``tokenize_phrase``, ``untokenize_word_vector``, ``__len__``), plus
``query_batch`` for batched serving (left-padded like word_utils.py:76-80).  Image and mask
decoding (cv2, scipy.io) is outside the hot path's scope.
"""
import os.path as osp

import torch

from .text import Corpus, Dictionary

# pickled class paths of the vocabulary objects -> the classes they are rebuilt as
_REF_MODULES = ("dmn_pytorch.utils.word_utils", "dmn_pytorch.utils", "utils.word_utils", "word_utils")
_SAFE = [(cls, "%s.%s" % (mod, cls.__name__)) for mod in _REF_MODULES for cls in (Corpus, Dictionary)]

SUPPORTED_DATASETS = {                                     # referit_loader.py:37-51
    'referit': ('train', 'val', 'trainval', 'test'),
    'unc': ('train', 'val', 'trainval', 'testA', 'testB'),
    'unc+': ('train', 'val', 'trainval', 'testA', 'testB'),
    'gref': ('train', 'val'),
}


def _safe_load(path):
    with torch.serialization.safe_globals(_SAFE):
        return torch.load(path, map_location="cpu", weights_only=True)


def load_corpus(path):
    """``corpus.pth`` -> ``dms_amd.text.Corpus`` with the file's vocabulary and ``max_len``."""
    obj = _safe_load(path)
    if not isinstance(obj, Corpus) or not isinstance(getattr(obj, "dictionary", None), Dictionary):
        raise ValueError("%s does not hold a word_utils.Corpus (got %s)" % (path, type(obj).__name__))
    d = obj.dictionary
    w2i, i2w = getattr(d, "word2idx", None), getattr(d, "idx2word", None)
    if not isinstance(w2i, dict) or not isinstance(i2w, list):
        raise ValueError("%s: Corpus.dictionary lacks word2idx / idx2word" % path)
    if len(w2i) != len(i2w) or any(not isinstance(w, str) or w2i.get(w) != i for i, w in enumerate(i2w)):
        raise ValueError("%s: word2idx and idx2word disagree" % path)
    if not hasattr(obj, "max_len"):
        obj.max_len = 20                                   # word_utils.py:47
    return obj


def load_split(path):
    """``<dataset>_<split>.pth`` -> list of ``(image file, mask file, phrase)`` string triples."""
    obj = _safe_load(path)
    if not isinstance(obj, (list, tuple)):
        raise ValueError("%s does not hold a list (got %s)" % (path, type(obj).__name__))
    out = []
    for i, item in enumerate(obj):
        if (not isinstance(item, (list, tuple)) or len(item) != 3
                or not all(isinstance(s, str) for s in item)):
            raise ValueError("%s: entry %d is not an (image, mask, phrase) string triple" % (path, i))
        out.append(tuple(item))
    return out


class QuerySet(object):
    """The query side of ``ReferDataset(data_root, split_root, dataset, split=..., max_query_len=...)``."""

    def __init__(self, split_root='data', dataset='referit', split='train', max_query_len=20):
        if dataset not in SUPPORTED_DATASETS:
            raise ValueError('Dataset {0} is not supported by this loader'.format(dataset))
        if split not in SUPPORTED_DATASETS[dataset]:                         # referit_loader.py:84-87
            raise ValueError('Dataset {0} does not have split {1}'.format(dataset, split))
        self.split_root, self.dataset, self.split = split_root, dataset, split
        self.query_len = max_query_len
        dataset_path = osp.join(split_root, dataset)
        self.corpus = load_corpus(osp.join(dataset_path, 'corpus.pth'))      # referit_loader.py:89
        splits = [split]
        if dataset != 'referit':                                             # referit_loader.py:91-93
            splits = ['train', 'val'] if split == 'trainval' else [split]
        self.images = []
        for s in splits:                                                     # referit_loader.py:94-97
            self.images += load_split(osp.join(dataset_path, '{0}_{1}.pth'.format(dataset, s)))

    def __len__(self):
        return len(self.images)

    def tokenize_phrase(self, phrase):
        return self.corpus.tokenize(phrase, self.query_len)

    def untokenize_word_vector(self, words):
        return self.corpus.dictionary[words]

    def phrase(self, idx):
        return self.images[idx][2]

    def query_batch(self, indices, device=None):
        """(len(indices), query_len) int64 ids of the phrases at `indices`, left-padded."""
        return self.corpus.tokenize_batch([self.images[i][2] for i in indices], self.query_len, device)

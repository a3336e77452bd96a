# -*- coding: utf-8 -*-
"""Query tokenisation with the reference's exact semantics (host side, feeds ``lang``).

Mirrors ``dmn_pytorch/utils/word_utils.py`` (``Dictionary`` :16-41, ``Corpus`` :44-94) --
same class and method names, so ``corpus.pth``-style vocabularies rebuilt through
``add_word`` give the same ids.  Quirks that matter for parity and are kept on purpose:

* the split pattern ``(\\W+)`` CAPTURES the separators (word_utils.py:13,70): blanks and
  punctuation are tokens of their own; the vocabulary is built from ``line.split()``
  (word_utils.py:62), so they normally map to ``<unk>``;
* a final ``'.'`` token is dropped (word_utils.py:73-74);
* queries are LEFT-padded with ``<pad>`` to ``max_len`` and truncated on the right
  (word_utils.py:76-80); the language SRU runs over the pads like over any word
  (dmn.py:147-156), which ``tokenize_batch`` therefore reproduces for batched serving.
"""
import codecs
import re

import torch

UNK_TOKEN = '<unk>'
PAD_TOKEN = '<pad>'
_SPLIT = re.compile(r'(\W+)')


class Dictionary(object):
    def __init__(self):
        self.word2idx = {}
        self.idx2word = []

    def add_word(self, word):
        idx = self.word2idx.get(word)
        if idx is None:
            idx = len(self.idx2word)
            self.idx2word.append(word)
            self.word2idx[word] = idx
        return idx

    def __len__(self):
        return len(self.idx2word)

    def __getitem__(self, a):
        if isinstance(a, int):
            return self.idx2word[a]
        if isinstance(a, list):
            return [self.idx2word[x] for x in a]
        if isinstance(a, str):
            return self.word2idx[a]
        raise TypeError("Query word/index argument must be int or str")

    def __contains__(self, word):
        return word in self.word2idx


class Corpus(object):
    def __init__(self):
        self.dictionary = Dictionary()
        self.max_len = 20

    def set_max_len(self, value):
        self.max_len = value

    def load_file(self, filename):
        with codecs.open(filename, 'r', 'utf-8') as f:
            for line in f:
                self.add_to_corpus(line.strip())
        self.dictionary.add_word(UNK_TOKEN)
        self.dictionary.add_word(PAD_TOKEN)

    def add_to_corpus(self, line):
        for word in line.split():
            self.dictionary.add_word(word.lower())

    def words(self, line, max_len=20):
        """The token strings ``tokenize`` turns into ids."""
        words = [w.lower() for w in _SPLIT.split(line.strip()) if len(w) > 0]
        if not words:
            raise IndexError("empty query (the reference raises here too, word_utils.py:73)")
        if words[-1] == '.':
            words = words[:-1]
        if max_len > 0:
            if len(words) > max_len:
                words = words[:max_len]
            elif len(words) < max_len:
                words = [PAD_TOKEN] * (max_len - len(words)) + words
        return words

    def tokenize(self, line, max_len=20):
        d = self.dictionary
        return torch.tensor([d[w] if w in d else d[UNK_TOKEN] for w in self.words(line, max_len)],
                            dtype=torch.long)

    def tokenize_batch(self, lines, max_len=20, device=None):
        """(B, max_len) int64 batch of left-padded queries (max_len > 0)."""
        if max_len <= 0:
            raise ValueError("a batch needs a fixed max_len > 0")
        out = torch.stack([self.tokenize(l, max_len) for l in lines])
        return out.to(device) if device is not None else out

    def __len__(self):
        return len(self.dictionary)

# -*- coding: utf-8 -*-
"""dms_amd -- MI355X-native DMN hot path (drop-in for dmn_pytorch's DMN / BaseDMN).

Public surface mirrors the reference package (dmn_pytorch/__init__.py:11,
dmn_pytorch/models/__init__.py:11) plus the SRU modules that replace the third-party
``sru`` dependency.
"""
from .models import DMN, BaseDMN, UpsamplingModule
from .sru import SRU, SRUCell
from .text import Corpus, Dictionary
from .metrics import SegmentationEvaluator, threshold_iou
from .serving import GraphedDMN

VERSION_INFO = (0, 1, 0)
__version__ = ".".join(map(str, VERSION_INFO))

__all__ = ["DMN", "BaseDMN", "UpsamplingModule", "SRU", "SRUCell", "Corpus", "Dictionary",
           "SegmentationEvaluator", "threshold_iou", "GraphedDMN"]

# -*- coding: utf-8 -*-
"""Models: same export list as the reference's dmn_pytorch/models/__init__.py:11."""
from .dmn import BaseDMN, DMN, UpsamplingModule

__all__ = ["BaseDMN", "DMN", "UpsamplingModule"]

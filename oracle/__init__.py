"""oracle/ -- TEST INFRASTRUCTURE ONLY: CPU restatements of the reference hot path.

Nothing under dms_amd/ may import this package (tests/test_boundary.py enforces it).
"""

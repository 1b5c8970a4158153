"""Stand-in module named like the third-party wheel fast-edit-distance==1.2.2
(/root/reference/pyproject.toml:22), which is absent from this image and from
/root/reference.  TEST INFRASTRUCTURE ONLY: it lets oracle/make_golden.py import
the real blaze.read_assignment from /root/reference IN THE BUILD CONTAINER so the
logic AROUND the arithmetic (strand, shortcut, fold, UMI correction, trimming,
record format) can be captured as golden vectors.  It pins nothing about
sub_edit_distance itself (parity unpinned, see oracle_ed.c).  Never shipped in
the product, never on sys.path outside make_golden.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as _o  # noqa: E402

_TIE = int(os.environ.get("ORACLE_TIE_RULE", "0"))


def sub_edit_distance(str1, str2, max_ed=5):
    return _o.sub_edit_distance(str1, str2, max_ed, _TIE)


def edit_distance(str1, str2, max_ed=5):
    return _o.edit_distance(str1, str2, max_ed)

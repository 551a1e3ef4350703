#!/usr/bin/env python3
"""COCO compressed-RLE string <-> run lengths.  CONVENIENCE, NOT ON THE PRODUCT PATH, PARITY UNPINNED.

The reference decodes segmentation annotations with pycocotools (datasets/imagenetC.py:10,21: ``mask_utils.decode``); pycocotools is a
third-party dependency that is neither vendored in /root/reference nor installed in this image, so this restatement of its published
string format (common/maskApi.c, rleToString / rleFrString: 6-bit characters offset by 48, 5 payload bits + continuation bit 0x20,
sign-extended by bit 0x10 of the last group, every value after the third stored as a delta against the value two places back) could
not be checked against it.  controlvar_amd.preprocess therefore takes decoded masks or UNCOMPRESSED counts only; this script converts
annotation files once, for users who want to drop the pycocotools call:

    python tools/coco_rle_string.py anns.json > anns_uncompressed.json

Known answers in tests/test_preprocess.py are derived BY HAND from the format description above, not from pycocotools output."""
from __future__ import annotations

import json
import sys
from typing import Sequence


def rle_from_string(s) -> list:
    """COCO compressed RLE string -> run lengths.  pycocotools (common/maskApi.c, rleFrString; the dependency is not vendored
    in the reference and not installed here, so this codec is restated from the published algorithm - parity UNPINNED):
    6-bit characters offset by 48, 5 payload bits + continuation bit 0x20, sign-extended by bit 0x10 of the last group,
    and every value after the third is a delta against the value two places back."""
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, 1
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = c & 0x20
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_to_string(cnts: Sequence[int]) -> str:
    """inverse of rle_from_string (rleToString); used by the tests"""
    out = bytearray()
    for i, x in enumerate(cnts):
        x = int(x)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return out.decode('ascii')


if __name__ == '__main__':
    data = json.load(open(sys.argv[1]))

    def walk(o):
        if isinstance(o, dict):
            if 'counts' in o and 'size' in o and isinstance(o['counts'], str):
                o['counts'] = rle_from_string(o['counts'])
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(data)
    json.dump(data, sys.stdout)

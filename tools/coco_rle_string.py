#!/usr/bin/env python3
"""COCO compressed-RLE string -> uncompressed run lengths, for a whole annotation file:

    python tools/coco_rle_string.py anns.json > anns_uncompressed.json

The codec is controlvar_amd.preprocess.rle_from_string (the product decodes compressed strings itself since round 5, with a one-time
warning: pycocotools - the reference's decoder, datasets/imagenetC.py:10,21 - is neither vendored nor installed here, so the restated
format is UNPINNED; known answers in tests/test_preprocess.py are derived by hand)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from controlvar_amd.preprocess import rle_from_string, rle_to_string      # noqa: E402,F401


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

#!/usr/bin/env python
"""Unpack <tag>_record_files.b64 (tools/final_record.sh: every JSON / kernel-stats file of a record, gzip + base64 in one JSON
object) into a directory -- the files byte for byte as the GPU box wrote them (nothing of a record is rebuilt by
hand).
usage: python tools/final_record_unpack.py gpurun_out/<tag>_record_files.b64 profiles/"""
import base64
import gzip
import json
import os
import sys


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for name, blob in json.load(open(src)).items():
        data = gzip.decompress(base64.b64decode(blob))
        path = os.path.join(dst, os.path.basename(name))
        if os.path.exists(path) and open(path, "rb").read() == data:
            continue
        with open(path, "wb") as f:
            f.write(data)
        print("wrote", path, len(data))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

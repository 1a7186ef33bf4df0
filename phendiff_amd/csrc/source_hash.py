#!/usr/bin/env python3
"""sha256 over the library's sources (csrc/*.hip, csrc/*.h, csrc/*.sh, include/phendiff_hip.h): identifies the code a build /
a committed profile belongs to.  Standalone (no torch, no package import) so build.sh can run it on any box;
`phendiff_amd._lib.source_hash()` loads this same file.

    python3 source_hash.py                 -> hash of the whole source set
    python3 source_hash.py conv_igemm      -> hash of what ONE object depends on (its .hip + every header + the build script)
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _digest(files):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def source_hash() -> str:
    files = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".hip", ".h", ".sh")))
    files.append(os.path.join(HERE, "..", "..", "include", "phendiff_hip.h"))
    return _digest(files)


def object_hash(name: str, flags: str = "") -> str:
    files = [os.path.join(HERE, name + ".hip")]
    files += sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".sh")))
    files.append(os.path.join(HERE, "..", "..", "include", "phendiff_hip.h"))
    return hashlib.sha256((_digest(files) + "|" + flags).encode()).hexdigest()


if __name__ == "__main__":
    print(object_hash(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "") if len(sys.argv) > 1 else source_hash())

#!/usr/bin/env python3
"""Digests of the files the reference indexer writes and `salt` never reads (.R.pac, .R.rpac, .R.ann, .R.amb, .R.forward.bwt / .occ / .sa),
for the lambda fixture and every index stress case: runs the REAL reference indexer (oracle/_ref/salt-idx, built from /root/reference by
oracle/Makefile) and writes idx.unread.sha256 next to each fixture's idx.* files.  The files themselves (~1 MB per case) are not committed."""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "salt-idx")
SUFFIXES = (".R.pac", ".R.rpac", ".R.ann", ".R.amb", ".R.forward.bwt", ".R.forward.occ", ".R.forward.sa")
dirs = [os.path.join(ROOT, "tests", "golden", "lambda")] + sorted(os.path.join(ROOT, "tests", "golden", "index_cases", d)
                                                                   for d in os.listdir(os.path.join(ROOT, "tests", "golden", "index_cases")))
for d in dirs:
    with tempfile.TemporaryDirectory() as t:
        subprocess.run([REF, "-k", "19", os.path.join(d, "genome.fa"), os.path.join(d, "snps.txt"), os.path.join(t, "idx")], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(os.path.join(d, "idx.unread.sha256"), "w") as f:
            for s in SUFFIXES:
                f.write("%s %s\n" % (s, hashlib.sha256(open(os.path.join(t, "idx" + s), "rb").read()).hexdigest()))
    print(d)

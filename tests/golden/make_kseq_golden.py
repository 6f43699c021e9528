#!/usr/bin/env python3
"""Writes tests/golden/kseq/: small FASTA/FASTQ inputs (well-formed and malformed) and expected.tsv, the
records the REAL reference record reader (src/kseq.h, instantiated and looped like src/sketch.cpp:17,462-479)
returns for them, obtained from oracle/_ref/ref_driver kseq.  Run in the build container:
    make -C oracle ref && python3 tests/golden/make_kseq_golden.py"""
import gzip
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "kseq")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

CASES = {
    "empty.fa": b"",
    "header_only.fa": b">x",
    "header_nl.fa": b">x\n",
    "one_base.fa": b">x\nA",
    "two_records.fa": b">a first\nACGTACGTAC\nGTAC\n>b\nTTTT\nGG\n",
    "cr_only.fa": b">x\nA\r\nC\r\n",
    "crlf_multi.fa": b">x desc\r\nACGT\r\nAC\r\n>y\r\nGG\r\n",
    "lone_cr_line.fa": b">x\n\r\nAC\n",
    "blank_lines.fa": b"\n\n>a desc\n\nACGT\n\nAC\n>b\n>c\nTT",
    "leading_junk.fa": b"junk\n>a\nACGT\n",
    "junk_with_gt.fa": b"ju>nk more\nACGT\n>b\nCC\n",
    "gt_in_seq.fa": b">a\nAC\n>\nGG\n",
    "gt_midline.fa": b">a\nAC>GT\nTT\n",
    "plus_line_in_fasta.fa": b">a\nACGT\n+\nIIII\n>b\nGG\n",
    "at_line_in_fasta.fa": b">a\nACGT\n@b\nGG\n",
    "no_trailing_newline.fa": b">a\nACGT\n>b\nGGC",
    "tabs_in_header.fa": b">a\tb c\nACGT\n",
    "lowercase_n.fa": b">a\nacgtNNNNacgt\nnnACGT\n",
    "simple.fq": b"@r\nACGT\n+\nIIII\n@r2\nAC\n+r2\nII\n",
    "multiline.fq": b"@r\nACGT\nACGT\n+\nIIII\nJJJJ\n@r2\nAC\n+\nII\n",
    "qual_starts_with_at.fq": b"@r\nACGT\n+\n@III\n@r2\nAC\n+\n@@\n",
    "qual_short.fq": b"@r\nACGT\n+\nII\n@r2\nAC\n+\nII\n",
    "qual_long.fq": b"@r\nACGT\n+\nIIIIII\n@r2\nAC\n+\nII\n",
    "truncated_no_qual.fq": b"@r\nACGT\n+\n",
    "truncated_plus.fq": b"@r\nACGT\n+",
    "mixed.fq": b">fa\nACGT\n@fq\nGGCC\n+\nIIII\n>fa2\nTT\n",
    "crlf.fq": b"@r\r\nACGT\r\n+\r\nIIII\r\n",
}


def main():
    os.makedirs(OUT, exist_ok=True)
    names = []
    for name, data in CASES.items():
        open(os.path.join(OUT, name), "wb").write(data)
        names.append(name)
    # gzip'd inputs: one member, and two concatenated members (gzread continues into the second)
    with open(os.path.join(OUT, "two_records.fa.gz"), "wb") as f:   # mtime=0: reproducible bytes
        f.write(gzip.compress(CASES["two_records.fa"], mtime=0))
    names.append("two_records.fa.gz")
    with open(os.path.join(OUT, "two_members.fa.gz"), "wb") as f:
        f.write(gzip.compress(b">a\nACGT\nAC", mtime=0))
        f.write(gzip.compress(b"GT\n>b\nTTTT\n", mtime=0))
    names.append("two_members.fa.gz")
    out = subprocess.run([DRIVER, "kseq"] + names, cwd=OUT, check=True, stdout=subprocess.PIPE).stdout.decode()
    open(os.path.join(OUT, "expected.tsv"), "w").write(
        "# file\trecords\tbases\tfnv1a(sequence bytes)\tfnv1a(quality bytes, '~' where a record has none)\trecord end offsets...\n"
        "# produced by the reference's own kseq.h through oracle/_ref/ref_driver kseq (tests/golden/make_kseq_golden.py)\n" + out)
    print(out)
    # the reference's base coding table (src/common.h:27-37)
    bm = subprocess.run([DRIVER, "basemap"], check=True, stdout=subprocess.PIPE).stdout.decode()
    open(os.path.join(HERE, "basemap.txt"), "w").write(bm)
    # sizeof / offsetof of the structs the reference writes to disk (sketch.h, shuffle.h)
    lay = subprocess.run([DRIVER, "layout"], check=True, stdout=subprocess.PIPE).stdout.decode()
    open(os.path.join(HERE, "layout.txt"), "w").write(lay)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the dev
container only; /root/reference does not exist on the GPU box).

Everything collected here is DATA held by the reference's own tests -- input
files and expected outputs -- never source text:

  refs/      <- src/refs/*            (64x8 reftest JPEGs + PNG, src/tests.rs:131-142)
  parser/    <- src/file/test-images/ (16 JPEG + expected segment dumps,
                                       src/file/tests.rs:69-99; the three dumps
                                       over 400 KB are stored gzip-compressed)
  scan/      <- benches/scan.dat      (benches/bench.rs:9)
  huffman/   <- the two expected code listings inside the expect![[...]]
                blocks of src/huffman.rs:359-546 (snapshot values only)
"""
import gzip
import pathlib
import re
import shutil
import sys

REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = pathlib.Path(__file__).resolve().parent


def main():
    for sub in ("refs", "parser", "scan", "huffman"):
        (OUT / sub).mkdir(exist_ok=True)
    for f in (REF / "src/refs").iterdir():
        shutil.copy(f, OUT / "refs" / f.name)
    for f in (REF / "src/file/test-images").iterdir():
        if f.suffix == ".log" and f.stat().st_size > 400_000:
            with open(f, "rb") as src, gzip.GzipFile(OUT / "parser" / (f.name + ".gz"), "wb", 9, mtime=0) as dst:
                dst.write(src.read())
        else:
            shutil.copy(f, OUT / "parser" / f.name)
    shutil.copy(REF / "benches/scan.dat", OUT / "scan" / "scan.dat")

    text = (REF / "src/huffman.rs").read_text()
    blocks = re.findall(r'expect!\[\[r#"\n(.*?)"#\]\]', text, re.S)
    assert len(blocks) == 2, len(blocks)
    for name, blk in zip(("luma_dc.txt", "luma_ac.txt"), blocks):
        lines = [ln.strip() for ln in blk.splitlines() if ln.strip()]
        (OUT / "huffman" / name).write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

"""Dev helper: build several -D variants of the library in parallel.  usage: build_variants.py [--only=score_topk.hip,..] tag:DEF=1,DEF2=3 ..."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd import build as b

only = ()
args = []
for a in sys.argv[1:]:
    if a.startswith("--only="):
        only = tuple(a[7:].split(","))
    else:
        args.append(a)
if only:
    b.build()

def one(spec):
    tag, _, defs = spec.partition(":")
    return b.build_variant(tag, [d for d in defs.split(",") if d], only)  # "tag:-mllvm -some-flag" passes raw hipcc flags

with ThreadPoolExecutor(max_workers=3) as ex:
    for out in ex.map(one, args):
        print(out)

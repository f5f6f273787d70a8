#!/usr/bin/env python3
"""Re-wrap the prose paragraphs of a markdown file to <= 118 columns (tables, headings and blank lines are kept as they are).  usage: reflow_md.py FILE"""
import sys, textwrap
p = sys.argv[1]
out, para = [], []
def flush():
    global para
    if not para:
        return
    first = para[0]
    text = " ".join(l.strip() for l in para)
    ind = "  " if (first.startswith("* ") or first.startswith("  ")) else ""
    out.extend(textwrap.wrap(text, width=118, initial_indent=("  " if first.startswith("  ") else ""), subsequent_indent=ind, break_long_words=False, break_on_hyphens=False))
    para = []
for line in open(p).read().split("\n"):
    if line.startswith("|") or line.startswith("#") or line.strip() == "":
        flush(); out.append(line)
    elif line.startswith("* "):
        flush(); para = [line]
    else:
        para.append(line)
flush()
open(p, "w").write("\n".join(out))
s = "\n".join(out)
print(p, len(s.encode()), "bytes; longest prose line", max(len(l) for l in out if not l.startswith("|")))

#!/usr/bin/env python3
"""Re-wrap the prose paragraphs of a markdown file that hold a line longer than 120 characters (tables, headings, lists, code
and indented text are left alone).  Usage: wrap_md.py FILE [WIDTH=118]"""
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path, encoding="utf8").read().split("\n")
out, para, code = [], [], False


def flush():
    global para
    if para:
        if any(len(l) > 120 for l in para) and not any(l.startswith(("|", "#", "```", "* ", "- ", "  ", ">")) for l in para):
            out.extend(textwrap.wrap(" ".join(l.strip() for l in para), width=width, break_long_words=False, break_on_hyphens=False))
        else:
            out.extend(para)
        para = []


for l in lines:
    if l.startswith("```"):
        flush()
        code = not code
        out.append(l)
    elif code or l.strip() == "" or l.startswith("#"):
        flush()
        out.append(l)
    else:
        para.append(l)
flush()
open(path, "w", encoding="utf8").write("\n".join(out))
bad = [(i, len(l)) for i, l in enumerate(out, 1) if len(l) > 120]
print(path, "lines", len(out), "over 120:", bad)

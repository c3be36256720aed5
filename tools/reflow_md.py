#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a column limit (default 118): paragraphs, block quotes and list items;
tables, code fences and headings are left alone.   python tools/reflow_md.py DESIGN.md [width]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out, i, fence = [], 0, False


def flush(block, first, rest):
    text = " ".join(l.strip() for l in block)
    out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest,
                             break_long_words=False, break_on_hyphens=False))


while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        fence = not fence
    if fence or l.startswith(("```", "|", "#")) or not l.strip():
        out.append(l); i += 1; continue
    m = re.match(r"^(\s*)(\* |\d+\. |> )?", l)
    indent, marker = m.group(1), m.group(2) or ""
    first = indent + marker
    rest = indent + ("> " if marker == "> " else " " * len(marker))
    block = [l[len(first):]]
    i += 1
    while i < len(lines):
        n = lines[i]
        if not n.strip() or n.startswith(("```", "|", "#")) or re.match(r"^\s*(\* |\d+\. )", n):
            break
        if marker == "> ":
            if not n.startswith(">"):
                break
            block.append(n[1:]); i += 1; continue
        if n.startswith(">"):
            break
        block.append(n); i += 1
    flush(block, first, rest)
open(path, "w").write("\n".join(out))

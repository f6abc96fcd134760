"""Index of every entry point `include/dsea.h` declares, as the markdown block INTEGRATION.md section 4 carries between its
`<!-- abi-index:begin -->` / `<!-- abi-index:end -->` markers:

    python tools/abi_index.py            # print the block
    python tools/abi_index.py --write    # rewrite the block inside INTEGRATION.md

One row per function, in header order: the header section it stands in, the line of its declaration, the reference lines its
comment cites (file:line as written there) and the first sentence of that comment.  tests/test_abi_cpu.py holds the committed
block to this generator, so a new export cannot be added without its row."""
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HEADER = os.path.join(ROOT, "include", "dsea.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- abi-index:begin -->", "<!-- abi-index:end -->"

# declarations whose comment names them only in abbreviated form ("dsea_gmres_begin / step / end"): their comment by its opening words
_CYCLE, _STAGES = "/* dsea_gmres_cycle: ONE cycle", "/* the three stages of a cycle"
OVERRIDE = {"dsea_gmres_work_doubles": _CYCLE, "dsea_gmres_cycle": _CYCLE, "dsea_gmres_step": _STAGES, "dsea_gmres_end": _STAGES}

_SECTION = re.compile(r"/\* -{20,} *(.+?)\s*(\*/)?$")
_DECL = re.compile(r"^(?:int|void|const char \*|int64_t|size_t)\s*\**\s*(dsea_[a-z0-9_]+)\(")
_CITE = re.compile(r"(?:[A-Za-z_/0-9]+\.py|examples/[A-Za-z_/0-9.]+):[0-9][0-9,\- ]*[0-9]|(?:[A-Za-z_/0-9]+\.py):[0-9]+")


def _first_sentence(comment):
    text = " ".join(line.strip().lstrip("*").strip() for line in comment.splitlines())
    text = re.sub(r"^/\*+\s*", "", text)
    text = re.sub(r"\s*\*/\s*$", "", text)
    text = re.sub(r"\s+", " ", text).strip()
    m = re.search(r"(?<=[a-z0-9)\]])[.;:] (?=[A-Z(`])", text)
    if m and m.start() >= 24:
        text = text[:m.start() + 1]
    if len(text) > 150:
        text = text[:147].rsplit(" ", 1)[0] + " ..."
    return text.replace("|", "\\|")


def rows():
    lines = open(HEADER).read().splitlines()
    comments, decls, sections = [], [], []          # (first line, last line, text) / (line, name) / (line, title, comment index)
    no = 0
    while no < len(lines):
        line = lines[no]
        if line.lstrip().startswith("/*"):
            first = no
            while "*/" not in lines[no]:
                no += 1
            text = "\n".join(lines[first:no + 1])
            m = _SECTION.match(lines[first])
            if m:
                sections.append((first + 1, m.group(1).strip().rstrip("*/").strip(), len(comments)))
            comments.append((first + 1, no + 1, text))
        else:
            d = _DECL.match(line)
            if d:
                decls.append((no + 1, d.group(1)))
        no += 1
    out = []
    for at, name in decls:
        section = [t for (ln, t, _) in sections if ln < at]
        sec_idx = [c for (ln, _, c) in sections if ln < at]
        sec_start = [ln for (ln, _, _) in sections if ln < at]
        lo = sec_start[-1] if sec_start else 0
        before = [c for c in comments if lo and lo <= c[0] and c[1] < at]
        own = [c for c in before if c[1] == at - 1]
        naming = [c for c in before if re.search(r"\b%s\b" % name, c[2])]
        grouped = ""
        forced = [c for c in comments if name in OVERRIDE and c[2].startswith(OVERRIDE[name])]
        if forced:
            c = forced[0]
        elif own:
            c = own[-1]
        elif naming:
            c = naming[-1]
        elif before and comments[sec_idx[-1]] is before[-1] and not [1 for (ln, n) in decls if before[-1][1] < ln < at and
                                                                     [k for k in comments if k[1] == ln - 1]]:
            c = before[-1]                               # declared straight under the section comment
        else:
            c = None
            prev = [n for (ln, n) in decls if ln < at]
            grouped = "declared with `%s`" % prev[-1] if prev else ""
        text = c[2] if c else ""
        is_section = bool(c) and bool(sec_idx) and comments[sec_idx[-1]] is c
        if is_section:                                   # the section comment: drop its dashed title line
            text = "\n".join(text.splitlines()[1:]) or text
        cites = sorted(set(x.strip() for x in _CITE.findall(text)))
        where = "%d" % c[0] if c else "—"
        out.append((name, section[-1] if section else "preamble", at, ", ".join(cites), where,
                    _first_sentence(text) if text else grouped))
    return out


def block():
    rs = rows()
    text = [BEGIN,
            "| # | Entry point | `dsea.h` section : line | Comment at line | Reference lines that comment cites | It begins |",
            "|---|---|---|---|---|---|"]
    for k, (name, section, no, cites, where, first) in enumerate(rs, 1):
        text.append("| %d | `%s` | %s : %d | %s | %s | %s |" % (k, name, section, no, where, cites or "—", first or "—"))
    text.append(END)
    return "\n".join(text), len(rs)


if __name__ == "__main__":
    b, count = block()
    if "--write" in sys.argv:
        doc = open(DOC).read()
        i, j = doc.index(BEGIN), doc.index(END) + len(END)
        open(DOC, "w").write(doc[:i] + b + doc[j:])
        print("INTEGRATION.md: %d entry points" % count)
    else:
        print(b)

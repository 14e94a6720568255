"""The boundary documents cite the reference by file:line; this checks the citations of ``gpr.py`` in ``include/gprx.h`` and
``INTEGRATION.md`` against the reference's source (build container only: /root/reference does not travel to the GPU box).

Rule: a ``gpr.py:A-B`` (or ``gpr.py:A``) citation must lie inside the file, and when the text within two lines of it names a function
or class that the reference's ``gpr.py`` defines, at least one of the ranges cited there must overlap that definition.
"""

import ast
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/gpras/gpr.py"

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="/root/reference is only present in the build container")


def _definitions():
    with open(REF) as f:
        src = f.read()
    tree = ast.parse(src)
    spans = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            spans.setdefault(node.name, []).append((node.lineno, node.end_lineno))
    return spans, src.count("\n") + 1


def _citations(text):
    """[(line index, [(a, b), ...])] for every line holding 'gpr.py:...' (comma-separated ranges after one 'gpr.py:' included)."""
    out = []
    for i, line in enumerate(text.splitlines()):
        ranges = []
        for m in re.finditer(r"gpr\.py:((?:\d+(?:-\d+)?)(?:,\s*:?\d+(?:-\d+)?)*)", line):
            for part in re.split(r",\s*:?", m.group(1)):
                a, _, b = part.partition("-")
                ranges.append((int(a), int(b or a)))
        if ranges:
            out.append((i, ranges))
    return out


@pytest.mark.parametrize("path", ["include/gprx.h", "INTEGRATION.md"])
def test_cited_ranges_contain_the_functions_named_beside_them(path):
    spans, n_lines = _definitions()
    with open(os.path.join(ROOT, path)) as f:
        text = f.read()
    lines = text.splitlines()
    cites = _citations(text)
    assert cites, f"{path} cites no gpr.py lines at all"
    names = sorted((n for n in spans if len(n) > 4 and n != "__init__"), key=len, reverse=True)
    # prose names of the drivers count as naming them
    aliases = {"multi-start": "_optimize_multi_start", "differential-evolution": "_optimize_differential_evolutions",
               "DE objective": "_optimize_differential_evolutions", "DE population": "_optimize_differential_evolutions",
               "Adam driver": "_optimize_adam", "two-stage": "_optimize_two_stage", "three-stage": "_optimize_three_stage"}
    bad = []
    for i, ranges in cites:
        for a, b in ranges:
            if not (1 <= a <= b <= n_lines):
                bad.append(f"{path}:{i + 1}: gpr.py:{a}-{b} is outside the file ({n_lines} lines)")
        # names on the citing line itself decide; a citation that opens a line may belong to the name that ended the previous line
        ctx = lines[i]
        if re.match(r"^[\s*|`(]*gpr\.py:", ctx) and i > 0:
            ctx = lines[i - 1] + " " + ctx
        named = {name for name in names if re.search(r"(?<![A-Za-z0-9_])" + re.escape(name) + r"(?![A-Za-z0-9_])", ctx)}
        named |= {target for word, target in aliases.items() if word in ctx}
        for name in sorted(named):
            ok = any(a <= hi and lo <= b for (lo, hi) in spans[name] for (a, b) in ranges)
            if not ok:
                bad.append(f"{path}:{i + 1}: names {name} (defined at {spans[name]}) but cites {ranges}")
    assert not bad, "\n".join(bad)

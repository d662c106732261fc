#!/usr/bin/env python3
"""The reference's usage pattern (README.md of pedroegsilva/gofindthem: NewFinder -> AddExpression* -> ProcessText),
through the Python mirror of the MI355X engine.  Needs a HIP device.

    python examples/finder_quickstart.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gofindthem_amd.finder import Finder, GpuEngine, PyRegexpEngine  # noqa: E402

# finder.NewFinder(&finder.GpuEngine{}, &finder.RegexpEngine{}, caseSensitive=false)
f = Finder(GpuEngine(), PyRegexpEngine(), False)
f.AddExpressionWithTag('"breakfast" and ("coffee" or "tea") and not "decaf"', "drinks")
f.AddExpressionWithTag('INORD("grind" and "brew" and "pour")', "recipe order")
f.AddExpressionWithTag(r'r"[0-9]+ ?ml" and "water"', "quantities")

texts = [
    "Breakfast: grind the beans, brew with 250 ml of water, pour the COFFEE.",
    "Breakfast with decaf coffee only.",
    "Pour first, then brew, then grind - tea for breakfast, the wrong way round.",
]

# one document per call, exactly like the reference ...
for t in texts:
    hits = f.ProcessText(t)
    print("%-75s -> %s" % (t[:75], [(r.ExpresionIndex, r.Tag) for r in hits]))

# ... or the batch extension: one uint32 bitmap row per document (bit i = expression i)
bitmap = f.ProcessTexts(texts)
print("batch bitmap:", [hex(int(row[0])) for row in bitmap])

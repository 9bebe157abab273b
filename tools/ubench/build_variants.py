#!/usr/bin/env python3
"""Builds several -D variants of the library in parallel: build_variants.py name1:-DFLAG1,-DFLAG2 name2:... (lib/libxgate_hip_<name>.so)"""
import concurrent.futures, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
specs = [a.split(":", 1) for a in sys.argv[1:]]
def b(sp): return g.build_variant(sp[0], [f for f in sp[1].split(",") if f])
with concurrent.futures.ThreadPoolExecutor(4) as ex: print("\n".join(ex.map(b, specs)))

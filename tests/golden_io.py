"""Decode the JSON flat lists written by oracle/make_golden.py."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def dec(v):
    if isinstance(v, dict):
        if 't' in v:
            return tuple(dec(x) for x in v['t'])
        return complex(*v['c'])
    return v


def frontend_lists():
    with open(os.path.join(GOLDEN, 'frontend.json')) as f:
        raw = json.load(f)
    return {k: [dec(x) for x in l] for k, l in raw.items()}


def npz(name):
    return np.load(os.path.join(GOLDEN, name))


def frontend_lists_named(fname):
    with open(os.path.join(GOLDEN, fname)) as f:
        raw = json.load(f)
    return {k: (None if l is None else [dec(x) for x in l]) for k, l in raw.items()}

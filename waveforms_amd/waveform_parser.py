"""`wave_eval("gaussian(10) << 100 + ...")` -- text front-end.

The reference parses with an ANTLR4-generated parser for waveforms/Waveform.g4
(waveform_parser.py:26-315); ANTLR and Java are not needed here: this is a hand-written
lexer + precedence-climbing parser for the SAME grammar, including its precedence quirks:

  * binary operators, tightest first: `** ^`, `* /`, `+ -`, `<< >>`, all LEFT-associative
    (the grammar declares no <assoc=right>, so a ** b ** c == (a ** b) ** c);
  * unary minus is the LAST alternative of the left-recursive rule, hence binds loosest:
    `-a + b` is -(a + b) and `a * -b + c` is a * (-(b + c)), exactly as ANTLR resolves it;
  * `(x,)` / `(x, y)` are tuples, `[x, y]` lists, `pi`, `e`, `inf` constants, numbers may
    carry a `j` suffix, function calls take positional then keyword arguments.
The result is passed through `.simplify()` and cached, as in the reference.
"""
from __future__ import annotations

import re
from ast import literal_eval
from functools import lru_cache

from numpy import e, inf, pi

from . import multy_drag, waveform


class WaveformParseError(Exception):
    pass


_TOKEN = re.compile(r"""
    (?P<ws>[ \t\r\n]+)
  | (?P<num>(?:\d+(?:\.\d*)?|\.\d+)(?:[eE][+-]?\d+)?j?)
  | (?P<str>"[^"\r\n]*"|'[^'\r\n]*')
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>\*\*|<<|>>|[-+*/^()\[\],=])
""", re.X)

_CONSTANTS = {'pi': pi, 'e': e, 'inf': inf}
_BINARY = {'**': 5, '^': 5, '*': 4, '/': 4, '+': 3, '-': 3, '<<': 2, '>>': 2}
_UNARY_MINUS = 1

FUNCTIONS = [
    'D', 'chirp', 'const', 'cos', 'cosh', 'coshPulse', 'cosPulse', 'cut', 'drag',
    'drag_sin', 'drag_sinx', 'exp', 'gaussian', 'general_cosine', 'hanning', 'interp',
    'mixing', 'mollifier', 'one', 'poly', 'samplingPoints', 'sign', 'sin', 'sinc', 'sinh',
    'square', 'step', 't', 'zero'
]


def _tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise WaveformParseError(f"Syntax error at column {pos}: unexpected {text[pos]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind != 'ws':
            out.append((kind, m.group(kind)))
    out.append(('end', ''))
    return out


class _Parser:
    def __init__(self, text):
        self.toks = _tokenize(text)
        self.i = 0

    def peek(self, k=0):
        return self.toks[min(self.i + k, len(self.toks) - 1)]

    def take(self, value=None):
        kind, val = self.toks[self.i]
        if value is not None and val != value:
            raise WaveformParseError(f"Syntax error: expected {value!r}, found {val!r}")
        self.i += 1
        return kind, val

    # expression: precedence climbing over the grammar's alternatives
    def expression(self, min_prec=0):
        left = self.primary()
        while True:
            kind, val = self.peek()
            prec = _BINARY.get(val) if kind == 'op' else None
            if prec is None or prec < min_prec:
                return left
            self.take()
            right = self.expression(prec + 1)      # left-associative
            left = self.apply(val, left, right)

    @staticmethod
    def apply(op, a, b):
        if op in ('**', '^'):
            return a**b
        if op == '*':
            return a * b
        if op == '/':
            return a / b
        if op == '+':
            return a + b
        if op == '-':
            return a - b
        if op == '<<':
            return a << b
        return a >> b

    def primary(self):
        kind, val = self.peek()
        if kind == 'op' and val == '-':
            self.take()
            return -self.expression(_UNARY_MINUS)
        if kind == 'op' and val == '(':
            return self.paren_or_tuple()
        if kind == 'op' and val == '[':
            return self.list_()
        if kind == 'num':
            self.take()
            return literal_eval(val)
        if kind == 'str':
            self.take()
            return literal_eval(val)
        if kind == 'id':
            if self.peek(1) == ('op', '('):
                return self.call()
            self.take()
            if val in _CONSTANTS:
                return _CONSTANTS[val]
            raise WaveformParseError(f"Unknown identifier '{val}'")
        raise WaveformParseError(f"Syntax error: unexpected {val!r}")

    def paren_or_tuple(self):
        self.take('(')
        first = self.expression()
        if self.peek() == ('op', ')'):
            self.take()
            return first
        items = [first]
        while self.peek() == ('op', ','):
            self.take()
            if self.peek() == ('op', ')'):
                break
            items.append(self.expression())
        self.take(')')
        return tuple(items)

    def list_(self):
        self.take('[')
        items = []
        if self.peek() != ('op', ']'):
            items.append(self.expression())
            while self.peek() == ('op', ','):
                self.take()
                items.append(self.expression())
        self.take(']')
        return items

    def call(self):
        _, name = self.take()
        func = None
        for mod in (waveform, multy_drag):
            func = getattr(mod, name, None)
            if func is not None:
                break
        if func is None or name.startswith('_'):
            raise WaveformParseError(f"Unknown function '{name}'")
        self.take('(')
        args, kwargs = [], {}
        while self.peek() != ('op', ')'):
            if self.peek()[0] == 'id' and self.peek(1) == ('op', '='):
                _, key = self.take()
                self.take('=')
                kwargs[key] = self.expression()
            else:
                if kwargs:
                    raise WaveformParseError('positional argument follows keyword argument')
                args.append(self.expression())
            if self.peek() == ('op', ','):
                self.take()
                if self.peek() == ('op', ')'):
                    raise WaveformParseError("Syntax error: trailing ',' in call")
            elif self.peek() != ('op', ')'):
                raise WaveformParseError(f"Syntax error: unexpected {self.peek()[1]!r}")
        self.take(')')
        return func(*args, **kwargs)


def parse_waveform_expression(expr: str):
    p = _Parser(expr)
    if p.peek()[0] == 'id' and p.peek(1) == ('op', '=') and p.peek(2) != ('op', '='):
        raise WaveformParseError("Assignment expressions are not supported")
    try:
        result = p.expression()
        if p.peek()[0] != 'end':
            raise WaveformParseError(f"Syntax error: unexpected {p.peek()[1]!r}")
        if isinstance(result, (int, float, complex)):
            result = waveform.const(result)
        return result.simplify()
    except WaveformParseError:
        raise
    except Exception as exc:
        raise WaveformParseError(f"Failed to parse expression '{expr}': {exc}")


@lru_cache(maxsize=1024)
def wave_eval(expr: str):
    """Parse and evaluate a waveform expression; SyntaxError on failure
    (reference: waveforms/waveform_parser.py:296-315)."""
    try:
        return parse_waveform_expression(expr)
    except Exception as exc:
        raise SyntaxError(f"Failed to parse expression '{expr}': {exc}")

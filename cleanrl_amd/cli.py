"""Dataclass -> command line, mirroring what ``tyro.cli(Args)`` gives the reference scripts.

``tyro`` is not installed in this image, so the flag surface of the reference (``cleanrl/ppo.py:19-70``)
is rebuilt on ``argparse``: every dataclass field ``foo_bar`` becomes ``--foo-bar`` (``--foo_bar`` is
accepted too -- the reference's docs, tests and benchmark scripts use both spellings), booleans become
``--flag / --no-flag`` pairs, ``Optional``/``None``-defaulted fields accept a value, ``List[int]`` fields
take space-separated ints (``--device-ids 0 1``) and ``Literal`` fields become ``choices``.  Field
docstrings (the string literal under each field) become the help text.
"""
from __future__ import annotations

import argparse
import ast
import dataclasses
import inspect
import sys
import textwrap
import typing
from typing import Any, List, Optional, Sequence, Type, TypeVar

T = TypeVar("T")


def _field_docs(cls) -> dict:
    """Attribute docstrings: the bare string literal that follows each annotated field (base classes first)."""
    docs = {}
    for klass in reversed(cls.__mro__):
        if klass is not object and dataclasses.is_dataclass(klass):
            docs.update(_own_field_docs(klass))
    return docs


def _own_field_docs(cls) -> dict:
    docs = {}
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(cls)))
    except (OSError, TypeError):
        return docs
    body = tree.body[0].body
    for node, nxt in zip(body, body[1:]):
        if isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name) and isinstance(nxt, ast.Expr) \
                and isinstance(getattr(nxt, "value", None), ast.Constant) and isinstance(nxt.value.value, str):
            docs[node.target.id] = nxt.value.value
    return docs


def _unwrap_optional(tp):
    if typing.get_origin(tp) is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return args[0]
    return tp


def _strtobool(v: str) -> bool:
    """distutils.util.strtobool's vocabulary."""
    t = v.strip().lower()
    if t in ("y", "yes", "t", "true", "on", "1"):
        return True
    if t in ("n", "no", "f", "false", "off", "0"):
        return False
    raise argparse.ArgumentTypeError(f"invalid truth value {v!r}")


def _normalise(argv: Sequence[str]) -> List[str]:
    out = []
    for tok in argv:
        if tok.startswith("--") and len(tok) > 2:
            name, eq, val = tok.partition("=")
            tok = name.replace("_", "-") + eq + val
        out.append(tok)
    return out


def parse(cls: Type[T], argv: Optional[Sequence[str]] = None, description: Optional[str] = None) -> T:
    """``args = parse(Args)`` -- the drop-in for ``args = tyro.cli(Args)``."""
    hints = typing.get_type_hints(cls)
    docs = _field_docs(cls)
    parser = argparse.ArgumentParser(prog=None, description=description or cls.__doc__,
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for f in dataclasses.fields(cls):
        tp = _unwrap_optional(hints.get(f.name, str))
        flag = "--" + f.name.replace("_", "-")
        default = f.default if f.default is not dataclasses.MISSING else (
            f.default_factory() if f.default_factory is not dataclasses.MISSING else None)
        help_ = docs.get(f.name, "")
        origin = typing.get_origin(tp)
        if tp is bool:
            # `--flag`, `--no-flag` (tyro) and `--flag False` / `--flag true` (the strtobool flags of the argparse-based
            # reference scripts, e.g. ppo_pettingzoo_ma_atari.py:24-31)
            parser.add_argument(flag, dest=f.name, nargs="?", const=True, type=_strtobool, default=default, help=help_)
            parser.add_argument("--no-" + f.name.replace("_", "-"), dest=f.name, action="store_false",
                                help=argparse.SUPPRESS)
        elif origin in (list, List):
            (elem,) = typing.get_args(tp) or (str,)
            parser.add_argument(flag, dest=f.name, type=elem, nargs="*", default=default, help=help_)
        elif origin is typing.Literal:
            choices = list(typing.get_args(tp))
            parser.add_argument(flag, dest=f.name, type=type(choices[0]), choices=choices, default=default, help=help_)
        else:
            conv: Any = tp if tp in (int, float, str) else str
            parser.add_argument(flag, dest=f.name, type=conv, default=default, help=help_)
    ns = parser.parse_args(_normalise(sys.argv[1:] if argv is None else argv))
    return cls(**vars(ns))

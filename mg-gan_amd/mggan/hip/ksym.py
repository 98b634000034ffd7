"""Kernel symbols: the ONE spelling of a HIP kernel's name shared by bench.py's `roofline` block, the rocprofv3
summaries under profiles/ and the counter tables (tools/hbm_traffic.py, tools/mfma_util.py).

`short("_Z11attn_kernelILi16ELb1EEviPKf...kd")` -> "attn_kernel<16,true>": the function name with EVERY template
argument (integers, booleans, builtin types), without the parameter list.  Two instantiations never share a name.
Pure Python, no dependencies (the tools run it on the GPU box and in the build container alike)."""
import re

_BUILTIN = {"f": "float", "d": "double", "i": "int", "j": "unsigned", "l": "long", "m": "unsigned long", "x": "long long",
            "y": "unsigned long long", "b": "bool", "c": "char", "h": "unsigned char", "s": "short", "t": "unsigned short"}


def _template_args(rest):
    """rest starts with 'I': -> ([args], index behind the closing 'E') or None if it holds something this does not parse."""
    i, out = 1, []
    while i < len(rest) and rest[i] != "E":
        if rest[i] == "L":  # literal: L <type> <value> E
            m = re.match(r"L([a-z])(n?\d+)E", rest[i:])
            if not m:
                return None
            ty, val = m.group(1), m.group(2)
            if ty == "b":
                out.append("true" if val == "1" else "false")
            else:
                out.append(("-" + val[1:]) if val.startswith("n") else val)
            i += m.end()
        elif rest[i] in _BUILTIN:
            out.append(_BUILTIN[rest[i]])
            i += 1
        else:
            return None
    return out, i + 1


def short(sym):
    """Mangled kernel symbol (with or without the '.kd' suffix) -> name<template arguments>; anything that is not an
    Itanium-mangled function name is returned as is (minus '.kd')."""
    sym = sym.strip()
    if sym.endswith(".kd"):
        sym = sym[:-3]
    m = re.match(r"_Z(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1))
    base = sym[m.end():m.end() + n]
    rest = sym[m.end() + n:]
    if rest.startswith("I"):
        t = _template_args(rest)
        if t is not None and t[0]:
            return "{}<{}>".format(base, ",".join(t[0]))
    return base


def family(name):
    """attn_kernel<16,true> -> attn_kernel."""
    return name.split("<", 1)[0]


def parse_launch_log(text):
    """The string mggan_launch_log_read fills ("symbol:threads;symbol:threads") -> [(short name, threads)]."""
    out = []
    for item in text.split(";"):
        if not item:
            continue
        sym, _, threads = item.rpartition(":")
        out.append((short(sym), int(threads or 0)))
    return out


def primary(launches):
    """The kernel an entry's time is booked on: the launch with the most threads (the first of equals) -- an entry's other
    launches are its finalize / fold tails (one to a few dozen workgroups)."""
    best = None
    for name, threads in launches:
        if name == "timestamp_kernel":  # (the device-clock marks bench.py puts around an entry)
            continue
        if best is None or threads > best[1]:
            best = (name, threads)
    return best[0] if best else None

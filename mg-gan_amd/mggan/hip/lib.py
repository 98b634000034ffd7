"""Loader for libmggan_hip.so.  Argument types are derived from the C header
(include/mggan_hip.h), which is the single source of truth for the ABI."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MGGAN_HIP_LIB: another build of the same library, for A/B measurements of a kernel variant on one box)
LIB_PATH = os.environ.get("MGGAN_HIP_LIB") or os.path.join(_HERE, "libmggan_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "..", "include", "mggan_hip.h"))


class HipError(RuntimeError):
    pass


def _ctype(decl):
    d = decl.strip()
    if "*" in d:
        return ctypes.c_void_p
    if "mggan_stream_t" in d:
        return ctypes.c_void_p
    base = d.rsplit(" ", 1)[0] if " " in d else d
    base = base.replace("const", "").strip()
    return {"int": ctypes.c_int, "unsigned": ctypes.c_uint, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
            "size_t": ctypes.c_size_t, "long long": ctypes.c_longlong}[base]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(const char\*|int|size_t)\s+(mggan_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        res = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        out[name] = (res, argtypes)
    return out


# functions whose int return value is data, not a status code
_VALUE_RETURNING = {"mggan_version", "mggan_launch_log", "mggan_launch_log_read", "mggan_wgrad_splits", "mggan_lstm_prep_size", "mggan_cnn_bwd_grid", "mggan_cnn_grid", "mggan_comm_arena_bytes", "mggan_social_rows_grid", "mggan_social_rows_splits",
                    "mggan_scene_attention_grid", "mggan_conv1_tail_floats", "mggan_scene_attention_partial_floats",
                    "mggan_social_rows_partial_floats", "mggan_rccl_available"}

_lib = None


class _Lib:
    """Attribute access returns a checked wrapper: non-zero status -> HipError."""

    def __init__(self, cdll):
        self._c = cdll
        self._w = {}
        self.trace = None  # list of (name, args, start_event, end_event) while profiling is on
        self.marks = None  # {"names": entries to bracket, "buf": int64 device buffer, "calls": [(name, args)]} -- device-clock
        #                    marks around chosen entries, ON THE ENTRY'S STREAM: they can be captured into a HIP graph, so a
        #                    replay times those kernels in the regime the replay runs in (bench.py: roofline block)
        self.sink = None   # a list: every call appends (entry, [(kernel symbol, threads)]) while the launch log is on
        self._logbuf = ctypes.create_string_buffer(8192)
        self.decls = parse_header()
        for name, (res, argtypes) in self.decls.items():
            fn = getattr(cdll, name)  # AttributeError here == header/library mismatch
            fn.argtypes = argtypes
            fn.restype = res

    def launch_log(self, on):
        """Switch the library's launch log (include/mggan_hip.h: mggan_launch_log) on / off; it starts empty."""
        self._c.mggan_launch_log(1 if on else 0)

    def read_launches(self):
        """[(kernel symbol as tools/ and profiles/ spell it, threads)] of the launches since the last read."""
        from .ksym import parse_launch_log

        buf = self._logbuf
        self._c.mggan_launch_log_read(buf, len(buf))
        return parse_launch_log(buf.value.decode())

    def __getattr__(self, name):
        w = self._w.get(name)
        if w is not None:
            return w
        if name not in self.decls:
            raise AttributeError(name)
        res, _ = self.decls[name]
        fn = getattr(self._c, name)
        if res is not ctypes.c_int or name in _VALUE_RETURNING:
            w = fn
        else:
            last_error = self._c.mggan_last_error

            def w(*args, _fn=fn, _name=name):
                tr, mk = self.trace, self.marks
                if mk is not None and _name in mk["names"] and 16 * (len(mk["calls"]) + 1) <= mk["buf"].numel() * 8:
                    slot = mk["buf"].data_ptr() + 16 * len(mk["calls"])
                    self._c.mggan_launch_log(1)  # (empties the log: the unmarked calls before this one were not read)
                    self._c.mggan_timestamp(slot, args[-1])
                    rc = _fn(*args)
                    self._c.mggan_timestamp(slot + 8, args[-1])
                    # (the two marks go through the logged launch macro too: they are not launches of the entry)
                    mk["calls"].append((_name, args, [k for k in self.read_launches() if not k[0].startswith("timestamp_kernel")]))
                    if rc != 0:
                        raise HipError("{} failed ({}): {}".format(_name, rc, last_error().decode()))
                    return
                if tr is not None:
                    import torch

                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self._c.mggan_launch_log(1)  # (launches of value-returning / untraced helpers since the last read)
                rc = _fn(*args)
                if tr is not None:
                    e1.record()
                    tr.append((_name, args, e0, e1, self.read_launches()))
                elif self.sink is not None:
                    self.sink.append((_name, self.read_launches()))
                if rc != 0:
                    raise HipError("{} failed ({}): {}".format(_name, rc, last_error().decode()))

        self._w[name] = w
        return w


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(
                "libmggan_hip.so not found at {} -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mg-gan_amd/csrc`; there is no CPU fallback".format(LIB_PATH))
        # torch ships its own libamdhip64: it has to be in the process first so that this library binds to the
        # SAME HIP runtime as the tensors it is handed (loaded the other way round, /opt/rocm's runtime comes in
        # next to torch's and every launch fails with "no ROCm-capable device is detected")
        import torch  # noqa: F401

        _lib = _Lib(ctypes.CDLL(LIB_PATH))
    return _lib


class _LazyLib:
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = getattr(load(), name)
        if callable(v) and name.startswith("mggan_"):
            # an entry's wrapper is fixed for the life of the process (tracing / marks are read inside it): from the second
            # use on `lib.mggan_x` is a plain instance attribute -- ~130 C-ABI calls per eager iteration came through two
            # Python-level __getattr__ each
            self.__dict__[name] = v
        return v


def start_trace():
    """Bracket every C-ABI call with HIP events on the current stream (all launches go to torch's current
    stream, so torch.cuda.Event sees them)."""
    l = load()
    l.launch_log(True)
    l.trace = []


def stop_trace():
    """-> {entry: (calls, [ms per call], [args per call], [launches per call])}; launches = [(kernel symbol, threads)]
    from the library's launch log (mggan/hip/ksym.py spells the symbols as the rocprofv3 tables under profiles/ do)."""
    import torch

    l = load()
    tr, l.trace = l.trace, None
    l.launch_log(False)
    torch.cuda.synchronize()
    out = {}
    for name, args, e0, e1, launches in tr or []:
        c, t, a, k = out.get(name, (0, [], [], []))
        out[name] = (c + 1, t + [e0.elapsed_time(e1)], a + [args], k + [launches])
    return out


class launches_of:
    """with launches_of() as seen: ...   ->  seen = [(entry, [(kernel symbol, threads)])] of every C-ABI call inside the
    block, eager or under stream capture (the log is taken at launch time on the host)."""

    def __enter__(self):
        l = load()
        l.launch_log(True)
        l.sink = []
        return l.sink

    def __exit__(self, *exc):
        l = load()
        l.sink = None
        l.launch_log(False)
        return False


lib = _LazyLib()

"""ctypes binding of libuavx.so (include/uavx.h).  There is NO fallback: if the HIP library is
missing or no MI355X is visible, every entry point raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("UAVX_LIB") or os.path.join(CSRC, "libuavx.so")  # UAVX_LIB: A/B builds of the same library (tools/)

OBS_DIM = 10
UW_OBS_DIM = 4
MAX_AGENTS = 64
MAX_LEVELS = 16
BODY_DIM = 6
ABI_VERSION = 3
FLAG_DONE, FLAG_COLLIDED, FLAG_VEL_F32, FLAG_INACTIVE = 1, 2, 4, 32
F32, F64 = 0, 1

# every symbol include/uavx.h declares (tests check the built library exports each of them)
SYMBOLS = (
    "uavx_version", "uavx_build_info", "uavx_selftest", "uavx_strerror", "uavx_create", "uavx_destroy", "uavx_last_error", "uavx_num_envs",
    "uavx_num_agents", "uavx_set_config", "uavx_set_body_rule", "uavx_num_bodies", "uavx_get_bodies", "uavx_set_bodies",
    "uavx_set_curriculum", "uavx_set_env_levels", "uavx_get_env_levels", "uavx_set_prefetch", "uavx_reset", "uavx_step", "uavx_step_k", "uavx_observe", "uavx_get_state",
    "uavx_set_state", "uavx_set_position_mode", "uavx_get_position_mode", "uavx_set_state_f64", "uavx_get_state_f64",
    "uavx_get_metrics", "uavx_get_nonfinite", "uavx_snapshot_bytes", "uavx_save", "uavx_load", "uavx_step_ex", "uavx_get_episode_stats", "uavx_clear_episode_stats",
    "uavx_uw_create", "uavx_uw_destroy", "uavx_uw_last_error",
    "uavx_uw_reset", "uavx_uw_step", "uavx_uw_observe", "uavx_uw_get_state", "uavx_uw_set_state",
    "uavx_uw_step_ex", "uavx_uw_get_episode_stats", "uavx_uw_clear_episode_stats",
)


class Config(ctypes.Structure):
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double), ("max_speed", ctypes.c_double),
                ("max_acceleration", ctypes.c_double), ("collider_radius", ctypes.c_double),
                ("d_sense", ctypes.c_double), ("tau", ctypes.c_double), ("num_agents", ctypes.c_int32),
                ("num_bodies", ctypes.c_int32)]


class BodyRule(ctypes.Structure):  # uavx_body_rule
    _fields_ = [("speed", ctypes.c_double), ("period", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("seed", ctypes.c_uint64)]


class Level(ctypes.Structure):  # uavx_level
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double), ("collider_radius", ctypes.c_double),
                ("d_sense", ctypes.c_double), ("n_active", ctypes.c_int32), ("b_active", ctypes.c_int32)]


class UWConfig(ctypes.Structure):
    _fields_ = [("x_size", ctypes.c_double), ("y_size", ctypes.c_double), ("max_speed", ctypes.c_double),
                ("max_acceleration", ctypes.c_double), ("tau", ctypes.c_double)]


class StateView(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("loc", "vel", "tgt", "init_d", "prev_d", "flags", "counters")]


UWStateView = StateView  # same field list (uavx_uw_state_view)


class StateViewF64(ctypes.Structure):  # uavx_state_view_f64
    _fields_ = [(n, ctypes.c_void_p) for n in ("loc", "tgt", "init_d", "prev_d")]


POS_F32, POS_F64 = 0, 1

ACTION_CARTESIAN, ACTION_POLAR = 0, 1
RESET_NEVER, RESET_AGENT0_DONE, RESET_ALL_DONE = 0, 1, 2


class StepArgs(ctypes.Structure):  # uavx_step_args
    _fields_ = [("actions", ctypes.c_void_p), ("action_dtype", ctypes.c_int32), ("action_mode", ctypes.c_int32),
                ("evaluate", ctypes.c_int32), ("reset_policy", ctypes.c_int32), ("step_cap", ctypes.c_uint32),
                ("track_returns", ctypes.c_int32), ("seed", ctypes.c_uint64), ("obs", ctypes.c_void_p),
                ("rew", ctypes.c_void_p), ("done", ctypes.c_void_p), ("reset_mask", ctypes.c_void_p),
                ("ended", ctypes.c_void_p), ("truncated", ctypes.c_void_p), ("flags_mode", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


FLAGS_ARRAYS, FLAGS_IN_DONE = 0, 1


class UWStepArgs(ctypes.Structure):  # uavx_uw_step_args
    _fields_ = [("actions", ctypes.c_void_p), ("action_dtype", ctypes.c_int32), ("action_mode", ctypes.c_int32),
                ("auto_reset", ctypes.c_int32), ("step_cap", ctypes.c_uint32), ("track_returns", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("seed", ctypes.c_uint64), ("obs", ctypes.c_void_p),
                ("rew", ctypes.c_void_p), ("done", ctypes.c_void_p), ("info_distance", ctypes.c_void_p),
                ("reset_mask", ctypes.c_void_p), ("ended", ctypes.c_void_p), ("truncated", ctypes.c_void_p)]


_lib = None


def _code_only(text):
    """C / C++ source without comments and with every whitespace run outside a literal reduced to one blank: what the compiler
    sees, near enough.  String and character literals are kept byte for byte (a `//` inside one is not a comment), and a
    preprocessor directive keeps its line to itself (where a `#define` ends is code)."""
    import re
    NL, SP, TB = "\x00", "\x01", "\x02"
    out, code, i, n = [], [], 0, len(text)

    def flush():                                          # the code since the last literal (line ends still marked)
        if code:
            out.append("".join(code))
            code.clear()
    while i < n:
        c = text[i]
        if c in "\"'":                                    # literal: copy to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            flush()
            out.append(text[i:j + 1].replace(" ", SP).replace("\t", TB)); i = j + 1      # blanks of a literal are code
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":               # (a line comment ending in a backslash continues: not used here)
                i += 1
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            code.append(" ")
        else:
            code.append(NL if c == "\n" else c); i += 1
    flush()
    lines, directive = [], False
    for line in "".join(out).split(NL):
        body = re.sub(r"[ \t\r\f\v]+", " ", line).strip()
        if not body:
            continue
        starts = body.startswith("#")
        if starts or directive:                           # a directive (or the continuation of one): its own line
            lines.append(("\n" if starts else "") + body + ("" if body.endswith("\\") else "\n"))
            directive = body.endswith("\\")
        else:
            lines.append(body + " ")
    return re.sub(r" +", " ", "".join(lines)).strip().replace(SP, " ").replace(TB, "\t")


def build_flags():
    """What else decides the machine code: the Makefile (comments dropped) and the variables a caller may override it with."""
    mk = open(os.path.join(CSRC, "Makefile"), "r", encoding="utf-8", errors="replace").read()
    mk = "\n".join(l.split("#", 1)[0].rstrip() for l in mk.splitlines() if l.split("#", 1)[0].strip())
    env = ";".join(f"{k}={os.environ[k]}" for k in ("HIPCC", "ARCH", "HIPFLAGS") if k in os.environ)
    return mk + "\n" + env


def source_hash():
    """Identity of the kernels a measurement belongs to: sha256 over the CODE of csrc/*.hip, csrc/*.hpp and include/uavx.h --
    comments and whitespace left out, so that a reworded comment does not orphan a profile -- plus the build recipe (the
    Makefile without its comments and any HIPCC / ARCH / HIPFLAGS override in the environment: -ffp-contract and the -D
    tuning knobs decide bit-exactness and speed as much as the sources do); first 16 hex digits.  profiles/*_pmc_summary.json
    carry it; bench.py drops a summary whose hash is not the one of the tree; the library embeds it (uavx_build_info) and
    the loader rebuilds on mismatch."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "uavx.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    if os.path.exists(os.path.join(CSRC, "Makefile")):
        h.update(b"Makefile")
        h.update(build_flags().encode())
    return h.hexdigest()[:16]


def build(force=False):
    """hipcc build of csrc/ into csrc/libuavx.so (gfx950).  Cross-compiles without a GPU.  Several processes may
    get here at once (torchrun ranks on a fresh checkout): the build runs under an exclusive file lock into a
    temporary name and is renamed into place, so nobody ever maps a half-written library."""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(LIB_PATH) and _up_to_date():
                return LIB_PATH
            tmp = f"libuavx.so.tmp{os.getpid()}"
            proc = subprocess.run(["make", "-C", CSRC, "-B", f"OUT={tmp}", f"SRCHASH={source_hash()}"], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True)
            if proc.returncode != 0:
                try:
                    os.unlink(os.path.join(CSRC, tmp))
                except OSError:
                    pass
                raise RuntimeError(f"uavx: building {LIB_PATH} failed (make exit {proc.returncode}):\n{proc.stdout[-4000:]}")
            os.replace(os.path.join(CSRC, tmp), LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _up_to_date():
    """Is LIB_PATH the build of the sources in the tree?  By content: the library carries the hash of the sources it was
    built from (uavx_build_info); file times say nothing after a checkout or a snapshot copy.  A library built by a
    hand-run make (no hash) falls back to comparing file times."""
    import glob
    import re
    # read from the file, not through dlopen: a mapped library stays mapped, and a later CDLL of the rebuilt file under the
    # same path would hand back the old one
    with open(LIB_PATH, "rb") as f:
        mark = re.search(rb"UAVX_SRC_HASH=([0-9a-f]*)\0", f.read())
    if mark is None:
        return False          # built before the marker existed
    built_from = mark.group(1).decode()
    if built_from:
        return built_from == source_hash()
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + [
        os.path.join(os.path.dirname(_HERE), "include", "uavx.h"), os.path.join(CSRC, "Makefile")]
    return os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)


def load():
    global _lib
    if _lib is not None:
        return _lib
    override = bool(os.environ.get("UAVX_LIB"))
    if not os.path.exists(LIB_PATH) or (not override and not _up_to_date()):
        # not built yet (fresh checkout), or older than its sources (the ABI structs grow between versions: a stale library
        # with the same symbol names would be read with the wrong layout): compile the HIP library; this is a build, not a
        # fallback, and a compile error is raised as such instead of being reported as "not built".  An explicit UAVX_LIB
        # (A/B builds) is taken as it is and only has to pass the version check below.
        build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C {CSRC}` "
            "(or __graft_entry__.build()). There is no CPU fallback for the env step path.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
    L.uavx_version.restype = i32
    if L.uavx_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} speaks ABI version {L.uavx_version()}, this package binds version {ABI_VERSION}: "
                           f"rebuild it (`make -B -C {CSRC}`)")
    L.uavx_build_info.restype = ctypes.c_char_p      # (exists from ABI version 3 on: bound behind the version check)
    L.uavx_selftest.argtypes = [i32, ctypes.POINTER(ctypes.c_uint64)]
    L.uavx_strerror.restype = ctypes.c_char_p
    L.uavx_strerror.argtypes = [i32]
    L.uavx_create.argtypes = [ctypes.POINTER(Config), i64, i64, i32, ctypes.POINTER(vp)]
    L.uavx_destroy.argtypes = [vp]
    L.uavx_last_error.argtypes = [vp]
    L.uavx_last_error.restype = ctypes.c_char_p
    L.uavx_num_envs.argtypes = [vp]
    L.uavx_num_envs.restype = i64
    L.uavx_num_agents.argtypes = [vp]
    L.uavx_set_config.argtypes = [vp, ctypes.POINTER(Config)]
    L.uavx_set_body_rule.argtypes = [vp, ctypes.POINTER(BodyRule)]
    L.uavx_num_bodies.argtypes = [vp]
    L.uavx_get_bodies.argtypes = [vp, vp, vp]
    L.uavx_set_bodies.argtypes = [vp, vp, vp]
    L.uavx_set_curriculum.argtypes = [vp, ctypes.POINTER(Level), i32, i32, i32, vp]
    L.uavx_set_env_levels.argtypes = [vp, vp, vp]
    L.uavx_get_env_levels.argtypes = [vp, vp, vp]
    L.uavx_set_prefetch.argtypes = [vp, i32]
    L.uavx_reset.argtypes = [vp, vp, u64, vp, vp]
    L.uavx_step.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.uavx_step_k.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp]
    L.uavx_observe.argtypes = [vp, vp, vp]
    L.uavx_get_state.argtypes = [vp, ctypes.POINTER(StateView), vp]
    L.uavx_set_state.argtypes = [vp, ctypes.POINTER(StateView), vp]
    L.uavx_set_position_mode.argtypes = [vp, i32, vp]
    L.uavx_get_position_mode.argtypes = [vp]
    L.uavx_set_state_f64.argtypes = [vp, ctypes.POINTER(StateViewF64), vp]
    L.uavx_get_state_f64.argtypes = [vp, ctypes.POINTER(StateViewF64), vp]
    L.uavx_get_metrics.argtypes = [vp, vp, vp]
    L.uavx_get_nonfinite.argtypes = [vp, vp, vp]
    L.uavx_snapshot_bytes.argtypes = [vp]
    L.uavx_snapshot_bytes.restype = i64
    L.uavx_save.argtypes = [vp, vp, vp]
    L.uavx_load.argtypes = [vp, vp, vp]
    L.uavx_step_ex.argtypes = [vp, ctypes.POINTER(StepArgs), vp]
    L.uavx_get_episode_stats.argtypes = [vp, vp, vp, vp]
    L.uavx_clear_episode_stats.argtypes = [vp, vp]
    L.uavx_uw_create.argtypes = [ctypes.POINTER(UWConfig), i64, i64, i32, ctypes.POINTER(vp)]
    L.uavx_uw_destroy.argtypes = [vp]
    L.uavx_uw_last_error.argtypes = [vp]
    L.uavx_uw_last_error.restype = ctypes.c_char_p
    L.uavx_uw_reset.argtypes = [vp, vp, u64, vp, vp]
    L.uavx_uw_step.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.uavx_uw_observe.argtypes = [vp, vp, vp]
    L.uavx_uw_get_state.argtypes = [vp, ctypes.POINTER(UWStateView), vp]
    L.uavx_uw_set_state.argtypes = [vp, ctypes.POINTER(UWStateView), vp]
    L.uavx_uw_step_ex.argtypes = [vp, ctypes.POINTER(UWStepArgs), vp]
    L.uavx_uw_get_episode_stats.argtypes = [vp, vp, vp, vp]
    L.uavx_uw_clear_episode_stats.argtypes = [vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("uavx_version",):
            fn.restype = i32
    _lib = L
    return L


def check(rc, handle=None, uw=False):
    if rc == 0:
        return
    L = load()
    msg = L.uavx_strerror(rc).decode()
    if handle:
        detail = (L.uavx_uw_last_error if uw else L.uavx_last_error)(handle)
        if detail:
            msg += ": " + detail.decode()
    if rc == -1:
        raise ValueError("uavx: " + msg)
    raise RuntimeError("uavx: " + msg)

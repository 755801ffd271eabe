"""ctypes binding of libppgfront.so (the MI355X-native PPG-SLAM feature front-end).

This is plumbing for tests and the benchmark: it mirrors include/ppgfront.h one to one and adds
nothing of its own.  The library has no CPU path; loading fails loudly when the .so is missing and
`Context()` fails loudly when no gfx950 device is present.

The directory name contains a hyphen, so import it with `load()` from `__graft_entry__` or
`tests/conftest.py` (importlib by path) under the module name ``ppg_slam_amd``.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libppgfront.so")
DEFAULT_WEIGHTS = os.path.join(_HERE, "weights", "ppg_weights.bin")
HEADER = os.path.join(_HERE, "..", "include", "ppgfront.h")

PPG_OK, PPG_EINVAL, PPG_EHIP, PPG_EIO, PPG_ENOMEM, PPG_ECAP, PPG_ERANGE, PPG_ECOMM = 0, -1, -2, -3, -4, -5, -6, -7
SHUTTER_ID_BYTES = 128
T_FEAT, T_JUNC, T_HEAT_RAW, T_DESCMAP, T_SCORE, T_HEAT, T_HEAT_REFINED, T_HEAT_REMAPPED = range(8)
MATCH_PLAIN, MATCH_INIT = 0, 1


class Config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("K", C.c_float * 9), ("D", C.c_float * 4),
                ("fisheye", C.c_int), ("device_id", C.c_int), ("weights_path", C.c_char_p), ("max_batch", C.c_int),
                ("desc_dim", C.c_int), ("junction_thresh", C.c_float), ("junction_nms_radius", C.c_int),
                ("junction_max_num", C.c_uint), ("line_valid_thresh", C.c_float), ("line_valid_ratio", C.c_float),
                ("line_dist_thresh", C.c_float), ("heatmap_refine_sz", C.c_int), ("line_heatmap_thresh", C.c_float),
                ("line_inlier_rate", C.c_float), ("precision", C.c_int)]


class FrameOut(C.Structure):
    _fields_ = [("n", C.c_int), ("kp_xy", C.POINTER(C.c_int32)), ("kp_score", C.POINTER(C.c_float)),
                ("kp_un", C.POINTER(C.c_float)), ("kp_out", C.POINTER(C.c_uint8)), ("desc", C.POINTER(C.c_float)),
                ("n_edges", C.c_int), ("edge_idx", C.POINTER(C.c_uint32)), ("edge_score", C.POINTER(C.c_float)),
                ("conn_off", C.POINTER(C.c_int32)), ("conn_idx", C.POINTER(C.c_uint32)),
                ("coline_off", C.POINTER(C.c_int32)), ("coline_pairs", C.POINTER(C.c_uint32))]


class MapGraph(C.Structure):
    _fields_ = [("m", C.c_int), ("desc", C.c_void_p), ("proj", C.c_void_p), ("viewcos", C.c_void_p),
                ("in_view", C.c_void_p), ("bad", C.c_void_p), ("nobs", C.c_void_p), ("edge_off", C.c_void_p),
                ("edge_id", C.c_void_p), ("edge_other", C.c_void_p), ("edge_ok", C.c_void_p)]


_lib = None


def mapped_runtimes():
    """Paths of the HIP / HSA / RCCL runtime images this process maps, by kind (read from /proc/self/maps)."""
    out = {"amdhip64": set(), "hsa-runtime64": set(), "rccl": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                base = os.path.basename(path)
                for kind in out:
                    if base.startswith("lib" + kind + ".so"):
                        out[kind].add(path)
    except OSError:
        pass
    return {k: sorted(v) for k, v in out.items()}


def _torch_bundled(name):
    """Path of a runtime library that an installed PyTorch-ROCm wheel bundles (torch/lib/<name>), or None."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    p = os.path.join(os.path.dirname(spec.origin), "lib", name)
    return p if os.path.exists(p) else None


def _one_hip_runtime():
    """Makes sure the process ends up with ONE HIP runtime image, whatever the import order.

    libppgfront.so needs `libamdhip64.so.7` and finds the ROCm installation's copy through its RUNPATH.  A PyTorch-ROCm
    wheel bundles its own copy under torch/lib and asks for it by the unversioned name `libamdhip64.so`, which the
    dynamic loader does not match against an image loaded under the soname: importing torch AFTER this library therefore
    maps a second HIP runtime and a second HSA runtime beside the first, both opening the GPU (seen as device discovery
    failing now and then).  The other order is harmless: torch's image carries the soname `libamdhip64.so.7`, so this
    library binds to it.  Hence: if a HIP runtime is mapped already, nothing to do; else, if an installed torch bundles
    one, that image is loaded first (torch later finds the same file), unless PPG_HIP_RUNTIME=system asks for the ROCm
    installation's runtime (then torch must not be imported afterwards; Context() checks)."""
    if mapped_runtimes()["amdhip64"]:
        return
    if os.environ.get("PPG_HIP_RUNTIME", "") == "system":
        return
    p = _torch_bundled("libamdhip64.so")
    if p:
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def check_single_runtime():
    """Raises when two HIP runtime images are mapped: a state in which GPU discovery fails intermittently.  (Two HSA images
    alone are what a process under rocprofv3 looks like - the profiler's tool library maps the ROCm installation's copy
    before the program starts - and are not refused.)"""
    m = mapped_runtimes()
    for kind in ("amdhip64",):
        if len(m[kind]) > 1:
            raise RuntimeError("two %s runtime images are mapped in this process (%s): import torch BEFORE loading libppgfront.so, "
                               "or let ppg_slam_amd.lib() load the library (it binds to the runtime torch bundles); "
                               "with PPG_HIP_RUNTIME=system torch must not be imported at all" % (kind, ", ".join(m[kind])))


def lib():
    """Load libppgfront.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C ppg-slam_amd/csrc` "
                               "(__graft_entry__.build()); the front-end has no CPU path")
        _one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        check_single_runtime()
        L.ppg_create_matcher.restype = C.c_void_p
        L.ppg_create_matcher.argtypes = [C.POINTER(Config)]
        L.ppg_set_tunables.argtypes = [C.c_void_p, C.POINTER(Config)]
        L.ppg_shutter_create.restype = C.c_void_p
        L.ppg_shutter_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ppg_shutter_destroy.argtypes = [C.c_void_p]
        L.ppg_shutter_unique_id.argtypes = [C.c_void_p]
        L.ppg_shutter_release.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.ppg_shutter_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ppg_shutter_drain.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ppg_shutter_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        L.ppg_stereo_swap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.ppg_stereo_match.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ppg_stereo_match_start.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.ppg_stereo_match_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ppg_create.restype = C.c_void_p
        L.ppg_create.argtypes = [C.POINTER(Config)]
        L.ppg_destroy.argtypes = [C.c_void_p]
        L.ppg_last_error.restype = C.c_char_p
        L.ppg_last_error.argtypes = [C.c_void_p]
        L.ppg_stream.restype = C.c_void_p
        L.ppg_stream.argtypes = [C.c_void_p]
        L.ppg_tensor_elems.restype = C.c_size_t
        L.ppg_tensor_elems.argtypes = [C.c_void_p, C.c_int]
        L.ppg_grid_build.restype = C.c_void_p
        L.ppg_grid_build.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int]
        L.ppg_grid_free.argtypes = [C.c_void_p]
        L.ppg_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.ppg_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FrameOut)]
        L.ppg_inference.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ppg_match_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.POINTER(C.c_int)]
        L.ppg_match_prev_batch.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.ppg_submit_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.ppg_wait_batch.argtypes = [C.c_void_p, C.POINTER(FrameOut), C.c_void_p, C.c_void_p]
        L.ppg_profile.argtypes = [C.c_void_p, C.c_int]
        L.ppg_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ppg_profile_flops.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_config(width, height, max_batch=1, max_num=500, weights=DEFAULT_WEIGHTS, device_id=0, precision=0):
    """precision: 0 fp32 (parity path), 1 bf16 hi/lo split with three products, 2 plain bf16, 3 fp16 hi/lo split
    (3x3 convolutions)."""
    cfg = Config()
    lib().ppg_default_config(C.byref(cfg), width, height)
    cfg.precision = precision
    cfg.max_batch = max_batch
    cfg.junction_max_num = max_num
    cfg.device_id = device_id
    cfg._weights_keepalive = os.path.abspath(weights).encode()
    cfg.weights_path = cfg._weights_keepalive
    return cfg


class PPGError(RuntimeError):
    pass


def _frame_to_dict(o):
    n, e = o.n, o.n_edges
    d = dict(n=n, n_edges=e)
    d["xy"] = np.ctypeslib.as_array(o.kp_xy, (max(n, 1) * 2,))[:2 * n].reshape(n, 2).copy()
    d["score"] = np.ctypeslib.as_array(o.kp_score, (max(n, 1),))[:n].copy()
    d["un"] = np.ctypeslib.as_array(o.kp_un, (max(n, 1) * 2,))[:2 * n].reshape(n, 2).copy()
    d["out"] = np.ctypeslib.as_array(o.kp_out, (max(n, 1),))[:n].copy()
    d["desc"] = np.ctypeslib.as_array(o.desc, (max(n, 1) * 256,))[:n * 256].reshape(n, 256).copy()
    d["edge_idx"] = (np.ctypeslib.as_array(o.edge_idx, (e * 2,)).reshape(e, 2).copy() if e else np.zeros((0, 2), np.uint32))
    d["edge_score"] = np.ctypeslib.as_array(o.edge_score, (e,)).copy() if e else np.zeros(0, np.float32)
    d["conn_off"] = np.ctypeslib.as_array(o.conn_off, (n + 1,)).copy()
    nc = int(d["conn_off"][n])
    d["conn_idx"] = np.ctypeslib.as_array(o.conn_idx, (nc,)).copy() if nc else np.zeros(0, np.uint32)
    d["coline_off"] = np.ctypeslib.as_array(o.coline_off, (n + 1,)).copy()
    nl = int(d["coline_off"][n])
    d["coline_pairs"] = (np.ctypeslib.as_array(o.coline_pairs, (nl * 2,)).reshape(nl, 2).copy() if nl else np.zeros((0, 2), np.uint32))
    return d


class Context:
    """Owns one ppg_ctx (one GPU, one stream)."""

    def __init__(self, cfg, matcher_only=False):
        """matcher_only: ppg_create_matcher, a context without networks, frame buffers or warm-up (one per matching thread)."""
        self.cfg = cfg
        self.L = lib()
        check_single_runtime()
        self.h = (self.L.ppg_create_matcher if matcher_only else self.L.ppg_create)(C.byref(cfg))
        if not self.h:
            raise PPGError(("ppg_create_matcher" if matcher_only else "ppg_create") + " failed: " + self.L.ppg_last_error(None).decode())
        self.H, self.W = cfg.height, cfg.width

    def set_tunables(self, cfg):
        """ppg_set_tunables: the thresholds of cfg apply from the next frame on; returns the status code."""
        return self.L.ppg_set_tunables(C.c_void_p(self.h), C.byref(cfg))

    def close(self):
        if getattr(self, "h", None):
            self.L.ppg_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may be gone already
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PPGError(f"libppgfront error {rc}: " + self.L.ppg_last_error(C.c_void_p(self.h)).decode())

    @property
    def stream(self):
        return self.L.ppg_stream(C.c_void_p(self.h))

    # ---- PPGExtractor::run ----
    def extract(self, img):
        return self.extract_batch([img])[0]

    def extract_batch(self, imgs, to_dict=True):
        n = len(imgs)
        arrs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        for a in arrs:
            if a.shape != (self.H, self.W):
                raise ValueError("image shape mismatch")
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        outs = (FrameOut * n)()
        self._chk(self.L.ppg_extract_batch(C.c_void_p(self.h), ptrs, n, self.W, outs))
        return [_frame_to_dict(outs[i]) for i in range(n)] if to_dict else outs

    def extract_batch_device(self, dev_ptr, n, to_dict=True):
        outs = (FrameOut * n)()
        self._chk(self.L.ppg_extract_batch_device(C.c_void_p(self.h), C.c_void_p(dev_ptr), n, outs))
        return [_frame_to_dict(outs[i]) for i in range(n)] if to_dict else outs

    def submit_batch_device(self, dev_ptr, n, match_prev=False, nnratio=0.9):
        self._chk(self.L.ppg_submit_batch_device(C.c_void_p(self.h), C.c_void_p(dev_ptr), n, 1 if match_prev else 0,
                                                 C.c_float(nnratio)))

    def submit_batch(self, imgs, match_prev=False, nnratio=0.9):
        n = len(imgs)
        arrs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        for a in arrs:
            if a.shape != (self.H, self.W):
                raise ValueError("image shape mismatch")
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        self._chk(self.L.ppg_submit_batch(C.c_void_p(self.h), ptrs, n, self.W, 1 if match_prev else 0, C.c_float(nnratio)))

    def wait_batch(self, to_dict=True, want_matches=True):
        """Returns (frames, n_matches[n], matches12[n, max_num]) of the oldest submitted batch."""
        B, M = int(self.cfg.max_batch), int(self.cfg.junction_max_num)
        outs = (FrameOut * B)()
        m12 = np.full((B, M), -1, np.int32) if want_matches else None
        cnts = np.zeros(B, np.int32)
        n = self.L.ppg_wait_batch(C.c_void_p(self.h), outs, _p(m12), _p(cnts))
        if n <= 0:
            self._chk(n if n < 0 else PPG_EINVAL)
        frames = [_frame_to_dict(outs[i]) for i in range(n)] if to_dict else outs
        return frames, cnts[:n], (m12[:n] if want_matches else None), n

    # ---- stages ----
    def inference(self, imgs):
        a = np.ascontiguousarray(np.stack(imgs), np.uint8)
        self._chk(self.L.ppg_inference(C.c_void_p(self.h), _p(a), a.shape[0], 0))

    def inference_device(self, dev_ptr, n):
        self._chk(self.L.ppg_inference(C.c_void_p(self.h), C.c_void_p(dev_ptr), n, 1))

    def detect_keypoints(self, n=1):
        self._chk(self.L.ppg_detect_keypoints(C.c_void_p(self.h), n))

    def detect_lines(self, n=1):
        self._chk(self.L.ppg_detect_lines(C.c_void_p(self.h), n))

    def gen_descriptors(self, n=1):
        self._chk(self.L.ppg_gen_descriptors(C.c_void_p(self.h), n))

    def collect(self, n=1):
        outs = (FrameOut * n)()
        self._chk(self.L.ppg_collect(C.c_void_p(self.h), n, outs))
        return [_frame_to_dict(outs[i]) for i in range(n)]

    def read_tensor(self, which, frame=0):
        n = self.L.ppg_tensor_elems(C.c_void_p(self.h), which)
        a = np.empty(n, np.float32)
        self._chk(self.L.ppg_read_tensor(C.c_void_p(self.h), which, frame, _p(a), C.c_size_t(n)))
        h, w = self.H // 8, self.W // 8
        shape = {T_FEAT: (h, w, 128), T_JUNC: (h, w, 65), T_HEAT_RAW: (self.H, self.W, 2), T_DESCMAP: (h, w, 256),
                 T_SCORE: (self.H, self.W), T_HEAT: (self.H, self.W), T_HEAT_REFINED: (self.H, self.W),
                 T_HEAT_REMAPPED: (self.H, self.W)}[which]
        return a.reshape(shape)

    def write_tensor(self, which, arr, frame=0):
        a = np.ascontiguousarray(arr, np.float32).ravel()
        self._chk(self.L.ppg_write_tensor(C.c_void_p(self.h), which, frame, _p(a), C.c_size_t(a.size)))

    # ---- matching ----
    def dist_matrix(self, A, B):
        A = np.ascontiguousarray(A, np.float32)
        B = np.ascontiguousarray(B, np.float32)
        D = np.empty((A.shape[0], B.shape[0]), np.float32)
        self._chk(self.L.ppg_dist_matrix(C.c_void_p(self.h), _p(A), A.shape[0], _p(B), B.shape[0], A.shape[1] if A.ndim == 2 else 256, _p(D)))
        return D

    def match_nn(self, A, B, cand_off=None, cand_idx=None, skipB=None, th=0.7, ratio=0.9, mode=MATCH_PLAIN):
        A = np.ascontiguousarray(A, np.float32).reshape(-1, 256)
        B = np.ascontiguousarray(B, np.float32).reshape(-1, 256)
        nA, nB = A.shape[0], B.shape[0]
        if cand_off is not None:
            cand_off = np.ascontiguousarray(cand_off, np.int32)
            cand_idx = np.ascontiguousarray(cand_idx, np.int32)
            if cand_idx.size == 0:
                cand_idx = np.zeros(1, np.int32)
        if skipB is not None:
            skipB = np.ascontiguousarray(skipB, np.uint8)
        m = np.full(max(nA, 1), -1, np.int32)
        best = np.zeros(max(nA, 1), np.float32)
        second = np.zeros(max(nA, 1), np.float32)
        cnt = C.c_int(0)
        self._chk(self.L.ppg_match_nn(C.c_void_p(self.h), _p(A), nA, _p(B), nB, 256, _p(cand_off), _p(cand_idx), _p(skipB),
                                      C.c_float(th), C.c_float(ratio), mode, _p(m), _p(best), _p(second), C.byref(cnt)))
        return m[:nA], best[:nA], second[:nA], cnt.value

    def match_nn_noalloc(self, dA, nA, dB, nB, d_match12, d_best, d_second, d_count, th=0.7, ratio=0.9, mode=MATCH_PLAIN,
                         stream=None, dim=256):
        """ppg_match_nn_noalloc: every pointer argument is a device address (int); returns the status code."""
        return self.L.ppg_match_nn_noalloc(C.c_void_p(self.h), C.c_void_p(dA), nA, C.c_void_p(dB), nB, dim, C.c_float(th),
                                           C.c_float(ratio), mode, C.c_void_p(d_match12), C.c_void_p(d_best),
                                           C.c_void_p(d_second), C.c_void_p(d_count), C.c_void_p(stream))

    def last_error(self):
        return self.L.ppg_last_error(C.c_void_p(self.h)).decode()

    def search_for_initialization(self, nnratio, desc1, desc2, pos2, prev_matched, window):
        desc1 = np.ascontiguousarray(desc1, np.float32).reshape(-1, 256)
        desc2 = np.ascontiguousarray(desc2, np.float32).reshape(-1, 256)
        pos2 = np.ascontiguousarray(pos2, np.float32)
        prev = np.array(prev_matched, np.float32, order="C", copy=True)
        n1, n2 = desc1.shape[0], desc2.shape[0]
        m12 = np.full(max(n1, 1), -1, np.int32)
        cnt = C.c_int(0)
        self._chk(self.L.ppg_search_for_initialization(C.c_void_p(self.h), C.c_float(nnratio), _p(desc1), n1, _p(desc2), n2,
                                                       _p(pos2), _p(prev), int(window), _p(m12), C.byref(cnt)))
        return cnt.value, m12[:n1], prev

    def match_frames(self, fa, fb, n1, nnratio=0.9):
        m12 = np.full(max(n1, 1), -1, np.int32)
        cnt = C.c_int(0)
        self._chk(self.L.ppg_match_frames(C.c_void_p(self.h), fa, fb, C.c_float(nnratio), _p(m12), C.byref(cnt)))
        return cnt.value, m12[:n1]

    def match_prev_batch(self, n, nnratio=0.9):
        """Frame b of the previous extract call (F1) against frame b of the last one (F2), resident."""
        M = int(self.cfg.junction_max_num)
        m12 = np.full((n, M), -1, np.int32)
        cnts = np.zeros(n, np.int32)
        self._chk(self.L.ppg_match_prev_batch(C.c_void_p(self.h), n, C.c_float(nnratio), _p(m12), _p(cnts)))
        return cnts, m12

    def profile(self, enable=True):
        self._chk(self.L.ppg_profile(C.c_void_p(self.h), 2 if enable == 2 else (1 if enable else 0)))

    def profile_read(self):
        ms = (C.c_double * 12)()
        launches = (C.c_longlong * 12)()
        self._chk(self.L.ppg_profile_read(C.c_void_p(self.h), ms, launches))
        fl = (C.c_double * 12)()
        self._chk(self.L.ppg_profile_flops(C.c_void_p(self.h), fl))
        self.last_profile_flops = {k: fl[i] for i, k in enumerate(("conv3x3", "conv_other", "conv1a"))}
        names = ("conv3x3", "conv_other", "conv1a", "keypoints", "match", "host_wait", "host_tail", "host_submit", "host_finish", "host_tail_frames",
                 "lines", "descriptors")
        return {k: (ms[i], launches[i]) for i, k in enumerate(names)}

    def tail_stats(self):
        """(frames collected, frames whose line-graph tail ran on the host)."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._chk(self.L.ppg_tail_stats(C.c_void_p(self.h), C.byref(a), C.byref(b)))
        return a.value, b.value

    def check_in_frustum(self, pose12, pos, normal, min_dist, max_dist, limit):
        """Frame::CheckInFrustum for m points (host arithmetic); returns in_view, proj, depth, view_cos (nan when unset)."""
        pos = np.ascontiguousarray(pos, np.float32)
        normal = np.ascontiguousarray(normal, np.float32)
        dmin = np.ascontiguousarray(min_dist, np.float32)
        dmax = np.ascontiguousarray(max_dist, np.float32)
        pose = np.ascontiguousarray(pose12, np.float32)
        K = np.array([self.cfg.K[k] for k in range(9)], np.float32)
        m = pos.shape[0]
        g = self.L.ppg_grid_build(C.byref(self.cfg), None, 0)
        in_view = np.zeros(m, np.uint8)
        proj = np.zeros((m, 2), np.float32)
        depth = np.zeros(m, np.float32)
        vc = np.full(m, np.nan, np.float32)
        self.L.ppg_check_in_frustum.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4
        rc = self.L.ppg_check_in_frustum(C.c_void_p(g), _p(K), _p(pose), m, _p(pos), _p(normal), _p(dmin), _p(dmax),
                                         C.c_float(limit), _p(in_view), _p(proj), _p(depth), _p(vc))
        self.L.ppg_grid_free(C.c_void_p(g))
        self._chk(rc)
        return in_view, proj, depth, vc

    def match_mutual(self, A, B, th, cand12=None, cand21=None):
        """Mutual nearest neighbours (Matcher::SearchBySim3's agreement check); cand = (off, idx) per direction or None."""
        A = np.ascontiguousarray(A, np.float32)
        B = np.ascontiguousarray(B, np.float32)
        m12 = np.full(A.shape[0], -1, np.int32)
        n = C.c_int(0)
        o12 = i12 = o21 = i21 = None
        if cand12 is not None:
            o12, i12 = np.ascontiguousarray(cand12[0], np.int32), np.ascontiguousarray(cand12[1], np.int32)
            o21, i21 = np.ascontiguousarray(cand21[0], np.int32), np.ascontiguousarray(cand21[1], np.int32)
        self._chk(self.L.ppg_match_mutual(C.c_void_p(self.h), _p(A), A.shape[0], _p(B), B.shape[0], A.shape[1], _p(o12), _p(i12),
                                          _p(o21), _p(i21), C.c_float(th), _p(m12), C.byref(n)))
        return n.value, m12

    def distinctive_descriptors(self, desc, seg_off):
        """MapPoint::ComputeDistinctiveDescriptors for every segment of `desc` rows; returns BestIdx per segment."""
        desc = np.ascontiguousarray(desc, np.float32)
        seg = np.ascontiguousarray(seg_off, np.int32)
        best = np.zeros(max(len(seg) - 1, 0), np.int32)
        self._chk(self.L.ppg_distinctive_descriptors(C.c_void_p(self.h), _p(desc), _p(seg), len(seg) - 1, _p(best)))
        return best

    def features_in_area(self, pos, x, y, r):
        pos = np.ascontiguousarray(pos, np.float32)
        n = pos.shape[0]
        g = self.L.ppg_grid_build(C.byref(self.cfg), _p(pos), n)
        out = np.zeros(max(n, 1), np.int32)
        cnt = self.L.ppg_features_in_area(C.c_void_p(g), x, y, r, _p(out), n)
        self.L.ppg_grid_free(C.c_void_p(g))
        return out[:cnt].copy()

    def match_projected(self, pos2, desc2, desc1, uv, radius, th, ratio=0.0, row_ok=None, row_claims=None,
                        slot_blocked=None, e2_max=-1.0):
        """ppg_match_projected: the windowed search shared by Matcher::SearchByProjection / Fuse.  Row i (descriptor
        desc1[i], projected to uv[i], window radius[i]) searches the key points at pos2 (undistorted) with descriptors
        desc2.  Returns (n_matches, best_idx, best_dist, slot_blocked after the call or None)."""
        pos2 = np.ascontiguousarray(pos2, np.float32)
        desc2 = np.ascontiguousarray(desc2, np.float32)
        desc1 = np.ascontiguousarray(desc1, np.float32)
        uv = np.ascontiguousarray(uv, np.float32)
        n1, n2 = desc1.shape[0], desc2.shape[0]
        radius = np.ascontiguousarray(np.broadcast_to(np.asarray(radius, np.float32), (n1,)))
        keep = [pos2, desc2, desc1, uv, radius]
        a = ProjArgs()
        g = self.L.ppg_grid_build(C.byref(self.cfg), _p(pos2), n2)
        a.grid = g
        a.desc2, a.n2, a.desc1, a.n1 = desc2.ctypes.data, n2, desc1.ctypes.data, n1
        a.uv, a.radius = uv.ctypes.data, radius.ctypes.data
        for name, arr in (("row_ok", row_ok), ("row_claims", row_claims)):
            if arr is not None:
                arr = np.ascontiguousarray(arr, np.uint8)
                keep.append(arr)
                setattr(a, name, arr.ctypes.data)
        blocked = None
        if slot_blocked is not None:
            blocked = np.array(slot_blocked, np.uint8, copy=True)
            a.slot_blocked = blocked.ctypes.data
        a.th, a.ratio, a.e2_max = th, ratio, e2_max
        best_idx = np.full(max(n1, 1), -1, np.int32)
        best_dist = np.zeros(max(n1, 1), np.float32)
        cnt = C.c_int(0)
        try:
            self._chk(self.L.ppg_match_projected(C.c_void_p(self.h), C.byref(a), _p(best_idx), _p(best_dist), C.byref(cnt)))
        finally:
            self.L.ppg_grid_free(C.c_void_p(g))
        return cnt.value, best_idx[:n1], best_dist[:n1], blocked

    def extend_map_matches(self, nnratio, th, g, desc, pos_un, edge_idx, conn_off, conn_idx, tracked, frame_mp, frame_me):
        keep = {k: np.ascontiguousarray(v) for k, v in g.items()}
        mg = MapGraph()
        mg.m = keep["desc"].shape[0]
        for k in ("desc", "proj", "viewcos", "in_view", "bad", "nobs", "edge_off", "edge_id", "edge_other", "edge_ok"):
            setattr(mg, k, keep[k].ctypes.data)
        desc = np.ascontiguousarray(desc, np.float32)
        pos_un = np.ascontiguousarray(pos_un, np.float32)
        edge_idx = np.ascontiguousarray(edge_idx, np.uint32)
        conn_off = np.ascontiguousarray(conn_off, np.int32)
        conn_idx = np.ascontiguousarray(conn_idx, np.uint32)
        tracked = np.array(tracked, np.uint8, copy=True)
        frame_mp = np.array(frame_mp, np.int32, copy=True)
        frame_me = np.array(frame_me, np.int32, copy=True)
        cnt = C.c_int(0)
        self._chk(self.L.ppg_extend_map_matches(C.c_void_p(self.h), C.c_float(nnratio), C.c_float(th), C.byref(mg), _p(desc),
                                                _p(pos_un), desc.shape[0], _p(edge_idx), edge_idx.shape[0], _p(conn_off),
                                                _p(conn_idx), _p(tracked), _p(frame_mp), _p(frame_me), C.byref(cnt)))
        return cnt.value, tracked, frame_mp, frame_me


class ProjArgs(C.Structure):
    """ppg_proj_args (include/ppgfront.h)."""
    _fields_ = [("grid", C.c_void_p), ("desc2", C.c_void_p), ("n2", C.c_int), ("desc1", C.c_void_p), ("n1", C.c_int),
                ("uv", C.c_void_p), ("radius", C.c_void_p), ("row_ok", C.c_void_p), ("row_claims", C.c_void_p),
                ("slot_blocked", C.c_void_p), ("th", C.c_float), ("ratio", C.c_float), ("e2_max", C.c_float)]


class Vocabulary:
    """ppg_vocab: a DBoW3 binary vocabulary loaded by the library (host)."""

    def __init__(self, path):
        self.L_ = lib()
        self.L_.ppg_vocab_load.restype = C.c_void_p
        self.L_.ppg_vocab_load.argtypes = [C.c_char_p]
        self.L_.ppg_vocab_free.argtypes = [C.c_void_p]
        self.h = self.L_.ppg_vocab_load(os.path.abspath(path).encode())
        if not self.h:
            raise PPGError("ppg_vocab_load failed: " + self.L_.ppg_last_error(None).decode())
        v = [C.c_int(0) for _ in range(7)]
        self.L_.ppg_vocab_info(C.c_void_p(self.h), *[C.byref(x) for x in v])
        self.k, self.L, self.n_nodes, self.n_words, self.dim, self.scoring, self.weighting = [x.value for x in v]

    def node(self, i):
        par, w, wid, nc = C.c_int(0), C.c_double(0), C.c_int(0), C.c_int(0)
        ch, d = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        rc = self.L_.ppg_vocab_node(C.c_void_p(self.h), i, C.byref(par), C.byref(w), C.byref(wid), C.byref(nc), C.byref(ch), C.byref(d))
        if rc:
            raise PPGError("ppg_vocab_node failed")
        return (par.value, w.value, wid.value, [ch[k] for k in range(nc.value)],
                np.ctypeslib.as_array(d, (self.dim,)).copy())

    def bow_vectors(self, word_id, weight, node_id):
        n = len(word_id)
        word_id = np.ascontiguousarray(word_id, np.int32)
        weight = np.ascontiguousarray(weight, np.float64)
        node_id = np.ascontiguousarray(node_id, np.int32)
        bw, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        fn, fo, fi = np.zeros(max(n, 1), np.int32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.int32)
        nb, nf = C.c_int(0), C.c_int(0)
        rc = self.L_.ppg_bow_vectors(C.c_void_p(self.h), _p(word_id), _p(weight), _p(node_id), n, _p(bw), _p(bv), C.byref(nb),
                                     _p(fn), _p(fo), _p(fi), C.byref(nf))
        if rc:
            raise PPGError("ppg_bow_vectors failed")
        bow = [(int(bw[k]), float(bv[k])) for k in range(nb.value)]
        fv = [(int(fn[q]), [int(x) for x in fi[fo[q]:fo[q + 1]]]) for q in range(nf.value)]
        return bow, fv

    def close(self):
        if getattr(self, "h", None):
            self.L_.ppg_vocab_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bow_transform(ctx, voc, desc, levelsup=4):
    """ppg_bow_transform for host descriptors."""
    desc = np.ascontiguousarray(desc, np.float32)
    n = desc.shape[0]
    w, wt, nd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64), np.zeros(max(n, 1), np.int32)
    ctx._chk(ctx.L.ppg_bow_transform(C.c_void_p(ctx.h), C.c_void_p(voc.h), _p(desc), n, levelsup, _p(w), _p(wt), _p(nd)))
    return w[:n], wt[:n], nd[:n]


def bow_transform_batch(ctx, voc, n_frames, levelsup=4):
    """ppg_bow_transform_batch: the resident descriptors of the last extracted batch."""
    M = int(ctx.cfg.junction_max_num)
    w, wt, nd = np.zeros((n_frames, M), np.int32), np.zeros((n_frames, M), np.float64), np.zeros((n_frames, M), np.int32)
    ctx._chk(ctx.L.ppg_bow_transform_batch(C.c_void_p(ctx.h), C.c_void_p(voc.h), n_frames, levelsup, _p(w), _p(wt), _p(nd)))
    return w, wt, nd

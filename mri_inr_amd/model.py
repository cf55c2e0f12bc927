"""``ModulatedSiren`` -- host-side mirror of the reference's model interface, backed by libmsiren.

Drop-in for the class of the same name in the reference (src/networks/modulated_siren.py:346-457)
as used by its evaluation path (test_mod_siren.py:96-120, src/util/error.py:138,235):

    model = ModulatedSiren(**17 kwargs)          # same names, same meaning
    model.load_state_dict(sd)                    # same keys/shapes
    model.to(device); model.eval()
    out = model(tiles)                           # (B, O, O) float32 -> (B, S, S) float32

Arrays may be numpy arrays or torch tensors (CPU or ROCm device tensors, which are consumed and
produced in place through their ``data_ptr()``); the result has the type of the input.  All
arithmetic happens in hand-written gfx950 kernels behind the C ABI of ``include/msiren.h``; there
is no PyTorch or numpy compute on this path and no CPU fallback.
"""

from __future__ import annotations

import collections
import ctypes as C
import os

import numpy as np

from . import _lib
from . import synthetic

_ACT = {"sine": _lib.ACT_SINE, "morlet": _lib.ACT_MORLET}


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _device_index(device) -> int | None:
    """'cuda', 'cuda:1', 1, torch.device('cuda', 1) -> ordinal; 'cpu' -> None."""
    if device is None:
        return 0
    if isinstance(device, (int, np.integer)):
        return int(device)
    s = str(device)
    if s.startswith("cpu"):
        return None
    if s.startswith(("cuda", "hip")):
        return int(s.split(":")[1]) if ":" in s else 0
    raise ValueError(f"unknown device {device!r}")


class DeviceArray:
    """A float32 array in HBM owned through the C ABI (msiren_dev_alloc / msiren_dev_free)."""

    def __init__(self, model: "ModulatedSiren", shape):
        self.model = model
        self.shape = tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * 4
        p = C.c_void_p()
        _lib.check(model._lib.msiren_dev_alloc(model._h, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    def copy_from(self, host: np.ndarray):
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        _lib.check(self.model._lib.msiren_memcpy_h2d(self.model._h, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        _lib.check(self.model._lib.msiren_memcpy_d2h(self.model._h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr and self.model._h:
            self.model._lib.msiren_dev_free(self.model._h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedBlock:
    """Page-locked host memory from msiren_host_alloc, exposed to numpy through __array_interface__; goes back to its model's
    pool when the last array over it is gone (numpy keeps the exporting object alive as the array's base)."""

    def __init__(self, pool, ptr, nbytes, shape):
        self._pool, self.ptr, self.nbytes = pool, ptr, nbytes
        self.__array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self._pool._give_back(self.ptr, self.nbytes)
        except Exception:
            pass


class _PinnedPool:
    """Recycles page-locked blocks by size (hipHostMalloc costs ~100 us; the blocks of a steady loop are reused).  At most
    `keep` idle blocks per size are kept; the rest is freed.  `cap_bytes` bounds what is handed out at one time when the caller
    did not ask for page-locked memory explicitly (the default outputs of the host-pointer calls): beyond it array() returns
    None and the caller falls back to ordinary memory -- an application that keeps thousands of outputs alive does not pin them all.

    The pool does NOT reference its model (blocks reference the pool, the model references the pool: a back reference would be a
    cycle, and `del model` would free the GPU handle only at the next gc pass -- while any result array was alive, never).  It
    holds the library and a one-element cell with the model's handle, which the model empties when it destroys the handle.  Blocks
    are freed through msiren_host_free(NULL, ptr): nothing of a handle is touched from whatever thread the last array dies on."""

    def __init__(self, keep=8, cap_bytes=512 << 20):
        self._lib, self._cell = None, [None]          # set by attach()
        self._free, self._keep, self._cap, self._out = {}, keep, cap_bytes, 0

    def attach(self, lib, handle):
        self._lib, self._cell[0] = lib, handle

    def detach(self):
        """The handle is about to be destroyed: idle blocks go, blocks under live arrays stay valid and free themselves later."""
        self.drain()
        self._cell[0] = None

    def array(self, shape, strict=True):
        shape = tuple(int(x) for x in shape)
        nbytes = max(4, int(np.prod(shape, dtype=np.int64)) * 4)
        if not strict and self._out + nbytes > self._cap:
            return None
        lst = self._free.get(nbytes)
        if lst:
            ptr = lst.pop()
        else:
            if self._cell[0] is None:
                if strict:
                    raise _lib.MsirenError("the model has no device handle (destroyed or never created)")
                return None
            p = C.c_void_p()
            rc = self._lib.msiren_host_alloc(self._cell[0], nbytes, C.byref(p))
            if rc != 0 and not strict:
                return None
            _lib.check(rc)
            ptr = p.value
        self._out += nbytes
        return np.asarray(_PinnedBlock(self, ptr, nbytes, shape))

    def _give_back(self, ptr, nbytes):
        self._out -= nbytes
        lst = self._free.setdefault(nbytes, [])
        if len(lst) < self._keep and self._cell[0] is not None:
            lst.append(ptr)
        elif self._lib is not None:
            self._lib.msiren_host_free(None, ptr)

    def drain(self):
        for lst in self._free.values():
            while lst:
                self._lib.msiren_host_free(None, lst.pop())


class ModulatedSiren:
    """See module docstring.  Constructor signature: modulated_siren.py:349-368."""

    def __init__(self, dim_in, dim_hidden, dim_out, num_layers, latent_dim, w0, w0_initial, use_bias,
                 dropout, modulate, encoder_type, encoder_path, outer_patch_size, inner_patch_size,
                 siren_patch_size, device, activation, *, residual=False, precision="auto"):
        # attribute names as in the reference (:389-398)
        self.dim_in = int(dim_in)
        self.dim_hidden = int(dim_hidden)
        self.dim_out = int(dim_out)
        self.num_layers = int(num_layers)
        self.latent_dim = int(latent_dim)
        self.w0 = float(w0)
        self.w0_initial = float(w0_initial)
        self.use_bias = bool(use_bias)
        self.dropout = float(dropout)  # identity in eval mode; kept for signature parity
        self.modulate = modulate        # stored and never read, as in the reference (:397)
        self.encoder_type = encoder_type
        self.encoder_path = encoder_path
        self.outer_patch_size = int(outer_patch_size)
        self.inner_patch_size = int(inner_patch_size)
        self.siren_patch_size = int(siren_patch_size)
        self.activation = activation
        self.residual = bool(residual)
        self.precision = precision
        self.training = True
        if self.dim_in != 2:
            raise ValueError(f"dim_in must be 2 (the coordinate grid is a 2-D meshgrid), got {dim_in}")
        if self.dim_out != 1:
            raise ValueError(f"dim_out must be 1 (squeeze(2)+rearrange in the reference forward), got {dim_out}")
        if encoder_type == "vgg":
            raise NotImplementedError("encoder_type='vgg' (ablation encoder, src/networks/encoding/vgg.py) is out of "
                                      "scope of the MI355X path; use encoder_type='custom'")
        self._lib = None
        self._h = None
        self._pinned = _PinnedPool()       # page-locked host arrays (pinned_empty, pin_outputs)
        self._pin_outputs = True           # outputs of the host-pointer calls come from the pool (the kernels store into them in place)
        self._device = _device_index(device)
        self._committed = False
        # a fresh model has random weights, like a fresh nn.Module
        sd = synthetic.make_state_dict(seed=0, dim_hidden=self.dim_hidden, num_layers=self.num_layers,
                                       latent_dim=self.latent_dim, w0=self.w0,
                                       siren_patch_size=self.siren_patch_size, use_bias=self.use_bias,
                                       with_encoder=(self.outer_patch_size == 32))
        if encoder_type != "custom":
            # reference: no `encoder` attribute is created for other types (:252-262) -> forward fails
            sd = {k: v for k, v in sd.items() if not k.startswith("encoder.")}
        elif encoder_path is not None:
            sd.update(self._load_encoder_checkpoint(encoder_path))
        self._sd = collections.OrderedDict(sd)
        # the configuration's key set and shapes, fixed at construction: what _pull_tensors iterates over, so that keys a
        # trunk-only blob dropped come back when a later blob carries them
        self._shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in self._sd.items())
        self.grid = self._sd["grid"]
        if self._device is not None and _lib_device_available():
            self._ensure_handle()

    # ------------------------------------------------------------------ nn.Module-like protocol --
    def _load_encoder_checkpoint(self, path):
        """FixedEncoder: torch.load(path)["state_dict"] of a FixedAutoencoder (siren_encoder.py:544-549)."""
        from .weights import load_checkpoint

        raw = load_checkpoint(os.fspath(path))
        raw = raw["state_dict"] if "state_dict" in raw else raw
        out = {}
        for k, v in raw.items():
            if k.startswith("encoder."):
                out["encoder.encoder." + k] = np.ascontiguousarray(v, dtype=np.float32)
        return out

    def _config(self) -> _lib.MsirenConfig:
        if self.activation not in _ACT:
            # the reference treats anything but "morlet" as sine (:120-123)
            act = _lib.ACT_SINE
        else:
            act = _ACT[self.activation]
        cfg = _lib.MsirenConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.dim_in, cfg.dim_hidden, cfg.dim_out = self.dim_in, self.dim_hidden, self.dim_out
        cfg.num_layers, cfg.latent_dim = self.num_layers, self.latent_dim
        cfg.w0, cfg.w0_initial = self.w0, self.w0_initial
        cfg.use_bias = int(self.use_bias)
        cfg.activation = act
        cfg.outer_patch_size, cfg.inner_patch_size = self.outer_patch_size, self.inner_patch_size
        cfg.siren_patch_size = self.siren_patch_size
        cfg.residual = int(self.residual)
        # "auto": the split-fp16 trunk (fp32-equivalent accuracy, ~3x faster) wherever the library supports the
        # shape (H = 256, 2 <= L <= 11, no residual); the library itself falls back to the fp32 trunk otherwise
        cfg.precision = {"auto": _lib.PREC_F16X3, "fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "f16x3": _lib.PREC_F16X3,
                         "f16": _lib.PREC_F16, "fp16": _lib.PREC_F16}[self.precision]
        cfg.device = int(self._device or 0)
        return cfg

    def _ensure_handle(self):
        if self._h is not None:
            return
        if self._device is None:
            raise _lib.MsirenError("ModulatedSiren has no CPU path: move it to a gfx950 device with .to('cuda')")
        self._lib = _lib.load()
        h = C.c_void_p()
        cfg = self._config()
        _lib.check(self._lib.msiren_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._pinned.attach(self._lib, h)
        self._committed = False

    def _push_tensors(self):
        """msiren_set_tensor for every entry of the state_dict (no commit)."""
        self._ensure_handle()
        for k, v in self._sd.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            _lib.check(self._lib.msiren_set_tensor(self._h, k.encode(), a.ctypes.data, a.size))
        self._committed = False

    def _pull_tensors(self):
        """Refresh the host mirror from the tensors the handle holds (after msiren_broadcast_weights / _import): every
        key of the configuration the handle holds is (re)stored, keys it does not hold (a trunk-only source: no encoder)
        are dropped from ``state_dict()`` -- the same policy on the import and on the broadcast path."""
        self._ensure_handle()
        old, new = self._sd, collections.OrderedDict()
        for k, shape in self._shapes.items():
            a = np.empty(shape, dtype=np.float32)
            rc = self._lib.msiren_get_tensor(self._h, k.encode(), a.ctypes.data, a.size)
            if rc == _lib.E_STATE:  # the source did not hold it
                if k == "grid":     # (the library rebuilds a missing grid buffer; the mirror keeps its own)
                    new[k] = old[k]
                continue
            _lib.check(rc)
            new[k] = a
        self._sd = new
        self.grid = self._sd["grid"]

    def export_weights(self) -> np.ndarray:
        """The state_dict as ONE flat float32 blob (msiren_weights_export): the payload of the multi-GPU weight
        broadcast, and what a host ships when it moves the weights itself."""
        self._ensure_handle()
        if not self._committed:
            self._push_tensors()
        n = C.c_size_t()
        _lib.check(self._lib.msiren_weights_blob_size(self._h, C.byref(n)))
        blob = np.empty(n.value, dtype=np.float32)
        _lib.check(self._lib.msiren_weights_export(self._h, blob.ctypes.data, blob.size))
        return blob

    def import_weights(self, blob: np.ndarray):
        """blob -> this model's tensors (replacing them) -> commit: exactly what a receiving rank of
        msiren_broadcast_weights executes (msiren_weights_import).  Keys the blob does not carry (e.g. a trunk-only
        source without encoder) are dropped from ``state_dict()``, as they are on the device."""
        self._ensure_handle()
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _lib.check(self._lib.msiren_weights_import(self._h, blob.ctypes.data, blob.size))
        self._committed = True
        self._pull_tensors()
        return self

    def _ensure_committed(self):
        self._ensure_handle()
        if self._committed:
            return
        self._push_tensors()
        _lib.check(self._lib.msiren_commit_weights(self._h))
        self._committed = True

    def expected_keys(self):
        return list(self._sd.keys())

    def last_trunk_kernel(self) -> str:
        """Name of the trunk instance the handle's most recent trunk launch used (msiren_last_trunk_kernel)."""
        self._ensure_handle()
        buf = C.create_string_buffer(128)
        _lib.check(self._lib.msiren_last_trunk_kernel(self._h, buf))
        return buf.value.decode()

    def profile_kernels(self) -> list:
        """Per trunk instance since msiren_profile_enable(h, 1): name, launches, summed ms, coordinates evaluated."""
        self._ensure_handle()
        out, i = [], 0
        while True:
            buf, n, ms, co = C.create_string_buffer(128), C.c_int64(), C.c_double(), C.c_int64()
            if self._lib.msiren_profile_read_kernel(self._h, i, buf, C.byref(n), C.byref(ms), C.byref(co)) != 0:
                return out
            out.append(dict(kernel=buf.value.decode(), launches=n.value, ms_total=ms.value, coords=co.value))
            i += 1

    def state_dict(self):
        return collections.OrderedDict((k, np.array(v, copy=True)) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict: bool = True):
        """Same contract as nn.Module.load_state_dict: key set and shapes must match."""
        new = {}
        for k, v in state_dict.items():
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            new[k] = np.ascontiguousarray(v, dtype=np.float32)
        # the key set is the configuration's (fixed at construction), not whatever the mirror holds after a trunk-only import
        missing = [k for k in self._shapes if k not in new]
        unexpected = [k for k in new if k not in self._shapes]
        errs = []
        if strict and unexpected:
            errs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if strict and missing:
            errs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
        for k, v in new.items():
            if k in self._shapes and tuple(v.shape) != self._shapes[k]:
                errs.append(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                            f"the shape in current model is {self._shapes[k]}.")
        if errs:
            raise RuntimeError("Error(s) in loading state_dict for ModulatedSiren:\n\t" + "\n\t".join(errs))
        self._sd = collections.OrderedDict((k, new[k] if k in new else self._sd[k]) for k in self._shapes
                                           if k in new or k in self._sd)  # (construction order, whatever came back)
        self.grid = self._sd["grid"]
        self._committed = False
        if self._h is not None:
            self._ensure_committed()
        return collections.namedtuple("IncompatibleKeys", "missing_keys unexpected_keys")(missing, unexpected)

    def to(self, device):
        idx = _device_index(device)
        if idx is None:
            raise _lib.MsirenError("ModulatedSiren (MI355X build) has no CPU path; .to('cpu') is not supported")
        if idx != self._device and self._h is not None:
            self._destroy_handle()
        self._device = idx
        self._ensure_committed()
        return self

    def cuda(self, device=None):
        return self.to(0 if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        # the reference's train mode only switches on nn.Dropout(0.1) (:156); inference path only
        if mode:
            raise NotImplementedError("the MI355X path implements eval-mode inference only")
        self.training = False
        return self

    def parameters(self):
        return [v for k, v in self._sd.items() if k != "grid"]

    def _destroy_handle(self):
        if self._h is not None and self._lib is not None:
            self._pinned.detach()   # (blocks still under a live array stay allocated: their arrays outlive the handle)
            self._lib.msiren_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    # --------------------------------------------------------------------------------- forward --
    def _run(self, host_fn, dev_fn, x, in_tail, out_shape_fn, extra_null=0):
        """Common marshalling: numpy / torch-cpu -> host entry point; torch device tensor -> *_dev."""
        self._ensure_committed()
        S = self.siren_patch_size
        if _is_torch(x):
            import torch

            if x.is_cuda:
                if x.device.index != self._device:
                    raise ValueError(f"input is on cuda:{x.device.index}, model on cuda:{self._device}")
                xx = x.detach().to(torch.float32).contiguous()
                self._check_tail(tuple(xx.shape), in_tail)
                B = out_shape_fn(tuple(xx.shape))
                out = torch.empty((B, S, S), dtype=torch.float32, device=x.device)
                torch.cuda.current_stream(x.device).synchronize()
                args = [self._h, xx.data_ptr(), B, out.data_ptr()] + [None] * extra_null
                _lib.check(dev_fn(*args))
                _lib.check(self._lib.msiren_sync(self._h))
                return out
            res = self._run(host_fn, dev_fn, x.detach().cpu().numpy(), in_tail, out_shape_fn, extra_null)
            return torch.from_numpy(res)
        a = np.ascontiguousarray(x, dtype=np.float32)
        self._check_tail(a.shape, in_tail)
        B = out_shape_fn(a.shape)
        out = self._pinned.array((B, S, S), strict=False) if self._pin_outputs and B else None
        if out is None:
            out = np.empty((B, S, S), dtype=np.float32)
        args = [self._h, a.ctypes.data if a.size else None, B, out.ctypes.data if out.size else None] + [None] * extra_null
        _lib.check(host_fn(*args))
        return out

    @staticmethod
    def _check_tail(shape, tail):
        if len(shape) != len(tail) or any(t is not None and s != t for s, t in zip(shape, tail)):
            want = tuple("B" if t is None else t for t in tail)
            raise ValueError(f"expected input of shape {want}, got {tuple(shape)}")

    def forward(self, tiles):
        """tiles (B, O, O) -> (B, S, S).  Reference: modulated_siren.py:435-457."""
        if self.encoder_type != "custom":
            raise AttributeError("'Encoder' object has no attribute 'encoder'")  # as the reference fails
        O = self.outer_patch_size
        self._ensure_handle()
        return self._run(self._lib.msiren_forward_tiles, self._lib.msiren_forward_tiles_dev, tiles,
                         (None, O, O), lambda s: s[0])

    __call__ = forward

    def forward_latent(self, z):
        """latent (B, Z) -> (B, S, S): Modulator + SirenNet (modulated_siren.py:325-343, 215-233)."""
        self._ensure_handle()
        return self._run(self._lib.msiren_forward_latent, self._lib.msiren_forward_latent_dev, z,
                         (None, self.latent_dim), lambda s: s[0], extra_null=1)

    def forward_mods(self, mods):
        """mods (L, B, H) (or the Modulator's tuple of L arrays (B, H)) -> (B, S, S)."""
        if isinstance(mods, (tuple, list)):
            if _is_torch(mods[0]):
                import torch

                mods = torch.stack(list(mods), 0)
            else:
                mods = np.stack([np.asarray(m) for m in mods], 0)
        self._ensure_handle()
        return self._run(self._lib.msiren_forward_mods, self._lib.msiren_forward_mods_dev, mods,
                         (self.num_layers, None, self.dim_hidden), lambda s: s[1])

    # ---- the reference's sub-modules as callables: model.encoder(tiles), model.modulator(z), model.net(coords, mods) ----
    def _host_call(self, fn, x, in_tail, out_shape):
        self._ensure_committed()
        torch_in = _is_torch(x)
        a = x.detach().cpu().numpy() if torch_in else np.asarray(x)
        a = np.ascontiguousarray(a, dtype=np.float32)
        self._check_tail(a.shape, in_tail)
        out = np.empty(out_shape(a.shape), dtype=np.float32)
        _lib.check(fn(self._h, a.ctypes.data if a.size else None, a.shape[0], out.ctypes.data if out.size else None))
        if torch_in:
            import torch

            return torch.from_numpy(out)
        return out

    def encoder(self, tiles):
        """tiles (B, O, O) -> latent (B, Z): the reference's ``model.encoder(tiles)`` (modulated_siren.py:420, 282-301)."""
        if self.encoder_type != "custom":
            raise AttributeError("'Encoder' object has no attribute 'encoder'")  # as the reference fails
        self._ensure_handle()
        O = self.outer_patch_size
        return self._host_call(self._lib.msiren_encode_tiles, tiles, (None, O, O), lambda s: (s[0], self.latent_dim))

    def modulator(self, z):
        """latent (B, Z) -> tuple of num_layers arrays (B, H): the reference's ``model.modulator(z)`` (modulated_siren.py:416,
        325-343; it returns a tuple, hidden layer by hidden layer)."""
        self._ensure_handle()
        stacked = self._host_call(lambda h, a, B, o: self._lib.msiren_modulate(h, a, B, o), z, (None, self.latent_dim),
                                  lambda s: (self.num_layers, s[0], self.dim_hidden))
        # msiren_modulate's batch argument is the latent's first dimension; the output is (L, B, H)
        return tuple(stacked[l] for l in range(self.num_layers))

    def net(self, coords, mods):
        """``SirenNet.forward(coords, mods)`` (modulated_siren.py:215-233) -> (B, P, 1).  The trunk kernels evaluate the model's
        own coordinate grid (the only coordinates the reference ever passes, :447-450): ``coords`` must be None or that grid
        repeated over the batch."""
        if coords is not None:
            c = coords.detach().cpu().numpy() if _is_torch(coords) else np.asarray(coords)
            g = np.asarray(self.grid, dtype=np.float32)
            if c.shape[-2:] != g.shape or not np.array_equal(np.broadcast_to(g, c.shape), c.astype(np.float32)):
                raise ValueError("net(coords, mods): the trunk evaluates the model's own grid only (pass coords=None or model.grid repeated over the batch)")
        out = self.forward_mods(mods)
        return out.reshape(out.shape[0], -1, 1)

    def reconstruct(self, images):
        """images (n, Hh, Ww) or (Hh, Ww) -> (n, nV*I, nH*I): the whole slice pipeline of
        metrics_error (src/util/error.py:231-249) on the device."""
        self._ensure_committed()
        a = images.detach().cpu().numpy() if _is_torch(images) else np.asarray(images)
        a = np.ascontiguousarray(a, dtype=np.float32)
        single = a.ndim == 2
        if single:
            a = a[None]
        if a.ndim != 3:
            raise ValueError(f"expected (n, H, W) images, got {a.shape}")
        n, Hh, Ww = a.shape
        nv, nh = C.c_int32(), C.c_int32()
        _lib.check(self._lib.msiren_recon_shape(self._h, Hh, Ww, C.byref(nv), C.byref(nh)))
        I = self.inner_patch_size
        out = self._pinned.array((n, nv.value * I, nh.value * I), strict=False) if self._pin_outputs and n else None
        if out is None:
            out = np.empty((n, nv.value * I, nh.value * I), dtype=np.float32)
        _lib.check(self._lib.msiren_reconstruct_slices(self._h, a.ctypes.data, n, Hh, Ww, out.ctypes.data))
        res = out[0] if single else out
        if _is_torch(images):
            import torch

            return torch.from_numpy(res)
        return res

    # -------------------------------------------------------------------- low-level helpers ----
    # ---- page-locked host memory: the counterpart of torch's pin_memory() / DataLoader(pin_memory=True) ----
    def pinned_empty(self, shape) -> np.ndarray:
        """A float32 numpy array in page-locked memory (msiren_host_alloc): calls that are handed such arrays copy by DMA
        without staging and pipeline upload / kernels / download inside the call.  Recycled when the array is gone."""
        self._ensure_handle()
        return self._pinned.array(shape)

    def pin_outputs(self, on: bool = True):
        """Outputs of the host-pointer calls in page-locked arrays from a recycling pool -- the kernels store into them in place, no
        download.  On by default since round 5, bounded: at most 512 MB of such outputs alive at a time (beyond that, ordinary arrays).
        pin_outputs(False): ordinary numpy arrays always."""
        self._ensure_handle()
        self._pin_outputs = bool(on)
        return self

    def device_array(self, shape) -> DeviceArray:
        self._ensure_handle()
        return DeviceArray(self, shape)

    def sync(self):
        self._ensure_handle()
        _lib.check(self._lib.msiren_sync(self._h))

    def device_string(self) -> str:
        """"cuda:<index>" of the device the model lives on (what ``.to()`` takes)."""
        return f"cuda:{self._device}"

    def device_info(self) -> dict:
        self._ensure_handle()
        name = C.create_string_buffer(256)
        cus, mhz, hbm = C.c_int32(), C.c_int32(), C.c_uint64()
        _lib.check(self._lib.msiren_device_info(self._h, name, C.byref(cus), C.byref(mhz), C.byref(hbm)))
        pci = C.create_string_buffer(32)
        _lib.check(self._lib.msiren_device_pci(self._h, pci))
        return dict(name=name.value.decode(), compute_units=cus.value, clock_mhz=mhz.value, hbm_bytes=hbm.value,
                    pci_bus_id=pci.value.decode(), device_index=self._device)

    def flops_per_coord(self) -> float:
        H, L = self.dim_hidden, self.num_layers
        return float(2 * 2 * H + (L - 1) * 2 * H * H + 2 * H)


def _lib_device_available() -> bool:
    try:
        return _lib.device_count() > 0
    except Exception:
        return False

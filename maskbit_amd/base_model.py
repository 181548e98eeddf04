"""Model base for the HIP-backed MaskBit modules.

Mirrors the surface of the reference's ``BaseModel`` (modeling/modules/base_model.py:44-185):
``load_pretrained`` / ``save_pretrained`` / ``.device`` / ``.dtype`` / ``num_parameters`` with the
same checkpoint format (a flat ``torch.save``d state_dict, optional key-prefix renaming), so the
reference's drivers keep working.  Unlike the reference, a model here is a *spec-driven parameter
tree*: subclasses list ``(checkpoint key, shape, init)`` and the tree of (empty) container modules
is generated from the dotted names, which reproduces the reference's state_dict keys exactly
without mirroring its module classes.  Compute happens in libmaskbit_hip.so; the torch parameters
are the weight store the engine is (re)loaded from whenever they change.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, Optional, Tuple, Union

import torch


class _Node(torch.nn.Module):
    """Empty container; exists only to give parameters their dotted checkpoint names."""


ParamSpec = Tuple[str, Tuple[int, ...], str]          # (key, shape, init kind)


def _init_tensor(shape, kind: str) -> torch.Tensor:
    if kind == "normal":                               # trunc-normal(0.02) like the reference's Linear/Embedding init
        return torch.nn.init.trunc_normal_(torch.empty(shape), mean=0.0, std=0.02)
    if kind == "ones":
        return torch.ones(shape)
    if kind == "zeros":
        return torch.zeros(shape)
    if kind == "kaiming":                              # conv default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = 1.0 / max(fan_in, 1) ** 0.5
        return torch.empty(shape).uniform_(-bound, bound)
    raise ValueError(kind)


class BaseModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._engine = None            # ctypes handle (c_void_p)
        self._engine_key = None        # (device index, capacity)
        self._engine_sig = None        # weight signature the engine was loaded from

    # ------------------------------------------------------------------ parameter tree
    def _attach(self, key: str, tensor: torch.Tensor, buffer: bool = False) -> None:
        parts = key.split(".")
        mod: torch.nn.Module = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        if buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], torch.nn.Parameter(tensor))

    def _build(self, specs: Iterable[ParamSpec]) -> None:
        for key, shape, kind in specs:
            self._attach(key, _init_tensor(tuple(shape), kind))

    # ------------------------------------------------------------------ reference surface
    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            return p.dtype
        return torch.float32

    def num_parameters(self, only_trainable: bool = False, exclude_embeddings: bool = False) -> int:
        skip = {"class_emb.weight"} if exclude_embeddings else set()
        return sum(p.numel() for n, p in self.named_parameters() if n not in skip and (p.requires_grad or not only_trainable))

    def save_pretrained(self, save_directory: Union[str, os.PathLike], save_function: Optional[Callable] = None,
                        state_dict: Optional[Dict[str, torch.Tensor]] = None) -> None:
        if os.path.isfile(save_directory):
            print(f"Provided path ({save_directory}) should be a directory, not a file")
            return
        os.makedirs(save_directory, exist_ok=True)
        path = os.path.join(save_directory, "pytorch_model.bin")
        (save_function or torch.save)(self.state_dict() if state_dict is None else state_dict, path)
        print(f"Model weights saved in {path}")

    def load_pretrained(self, pretrained_model_path: Union[str, os.PathLike], strict_loading: bool = True,
                        torch_dtype: Optional[torch.dtype] = None, rename_keys: Optional[Dict[str, str]] = None) -> None:
        path = str(pretrained_model_path)
        if os.path.isdir(path):
            path = os.path.join(path, "pytorch_model.bin")
        if not os.path.isfile(path):
            raise ValueError(f"{path} does not exist")
        ckpt = torch.load(path, map_location="cpu")
        if rename_keys:
            renamed = {}
            for k, v in ckpt.items():
                for old, new in rename_keys.items():
                    if k.startswith(old):
                        k = k.replace(old, new)
                        break
                renamed[k] = v
            ckpt = renamed
        self.load_state_dict(ckpt, strict=strict_loading)
        if torch_dtype is not None:
            if not isinstance(torch_dtype, torch.dtype):
                raise ValueError(f"{torch_dtype} needs to be of type `torch.dtype`, e.g. `torch.float16`, but is {type(torch_dtype)}.")
            self.to(torch_dtype)
        self.eval()

    # ------------------------------------------------------------------ engine plumbing
    def _weight_signature(self):
        """Identity of the current weights (storage, in-place version counter, dtype per entry).  Computed once per engine() call -- the walk over ~300 entries costs
        ~1 ms of host time and engine() sits on the per-step-chunk path --: a caller may hand the value on (`_sig_hint`) for the duration of ONE call."""
        hint = getattr(self, "_sig_hint", None)
        if hint is not None:
            return hint
        return tuple((k, t.data_ptr(), t._version, t.dtype) for k, t in self.state_dict(keep_vars=True).items())

    def _require_cuda(self, what: str) -> torch.device:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__}.{what} runs only on an AMD GPU through libmaskbit_hip.so (model is on {dev}); "
                "maskbit_amd has no CPU path. Move the model with .to('cuda').")
        return dev

    def _drop_engine(self) -> None:
        if self._engine is not None:
            self._engine_destroy(self._engine)
            self._engine = None
            self._engine_key = None
            self._engine_sig = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    # subclasses provide: _engine_create(capacity) -> handle, _engine_destroy(handle), _engine_load(handle, key, tensor, stream)
    def _ensure_engine(self, capacity: int):
        """Create (or grow) the device engine and (re)load weights if the torch parameters changed."""
        dev = self._require_cuda("forward")
        key_dev = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is not None and (self._engine_key[0] != key_dev or self._engine_key[1] < capacity):
            self._drop_engine()
        with torch.cuda.device(key_dev):
            if self._engine is None:
                self._engine = self._engine_create(capacity)
                self._engine_key = (key_dev, capacity)
            sig = self._weight_signature()
            if sig != self._engine_sig:
                stream = torch.cuda.current_stream().cuda_stream
                keep = []
                for k, t in self.state_dict(keep_vars=True).items():
                    if not t.is_floating_point():
                        continue                                   # integer buffers are derived on the device
                    src = t.detach()
                    if src.dtype != torch.float32 or not src.is_contiguous():
                        src = src.float().contiguous()
                    keep.append(src)
                    self._engine_load(self._engine, k, src, stream)
                torch.cuda.current_stream().synchronize()          # temporaries in `keep` must outlive the repack
                self._engine_sig = sig
        return self._engine

"""Attention-backend selection by name: the flag the reference lacks.

The reference constructs its backend with a hard-wired line (engine/model_runner.py:90-91: one constructor call, the
other commented out) and `ServerArgs` (utils/args.py:14-141) has no field for it.  `select_backend(name, model_runner)`
is what that line becomes, and `add_backend_argument(parser)` / `BACKEND_CHOICES` are what `ServerArgs.add_cli_args`
gains next to `--page-size` (utils/args.py:27, 96-101).  INTEGRATION.md section 2 shows the three edits.

The reference's own two backends are imported lazily, by name, only when asked for: this package never imports
`minisglang` otherwise (its FA3 backend needs the CUDA-only `sgl_kernel` wheel, flash_attention_backend.py:5).
"""
from __future__ import annotations

import importlib
from typing import Callable, Dict, Tuple

# name -> (module, class); the first entry of every alias group is the canonical name
_REFERENCE_BACKENDS: Dict[str, Tuple[str, str]] = {
    "torch_native": ("minisglang.layers.attention_backends.torch_native_backend", "TorchNativeAttnBackend"),
    "fa3": ("minisglang.layers.attention_backends.flash_attention_backend", "FlashAttentionBackend"),
}
_ALIASES = {"mi355": "mi355", "mi355x": "mi355", "hip": "mi355",
            "torch_native": "torch_native", "torch-native": "torch_native", "native": "torch_native", "sdpa": "torch_native",
            "fa3": "fa3", "flash_attention": "fa3", "flash-attention": "fa3", "flashattention": "fa3"}

BACKEND_CHOICES = ("mi355", "torch_native", "fa3")
DEFAULT_BACKEND = "mi355"


def canonical_backend_name(name: str) -> str:
    """'mi355' | 'torch_native' | 'fa3' for any accepted spelling; ValueError (listing the choices) otherwise."""
    key = str(name).strip().lower()
    if key not in _ALIASES:
        raise ValueError(f"unknown attention backend {name!r}: choose one of {', '.join(BACKEND_CHOICES)}")
    return _ALIASES[key]


def backend_class(name: str) -> Callable:
    """The backend class for `name`.  'mi355' is this package's; the reference's are imported on demand and a missing
    dependency (no `minisglang` on sys.path, no `sgl_kernel`) surfaces as ImportError naming the backend - never as a
    silent fall-back to another backend."""
    canon = canonical_backend_name(name)
    if canon == "mi355":
        from .backend import MI355AttnBackend
        return MI355AttnBackend
    module, cls = _REFERENCE_BACKENDS[canon]
    try:
        return getattr(importlib.import_module(module), cls)
    except ImportError as e:
        raise ImportError(f"attention backend {canon!r} needs {module}.{cls}, which could not be imported: {e}") from e


def select_backend(name: str, model_runner):
    """What engine/model_runner.py:90-91 becomes: `self.attn_backend = select_backend(server_args.attention_backend, self)`.
    Every backend takes the model runner as its one constructor argument (torch_native_backend.py:16,
    flash_attention_backend.py:45, backend.py MI355AttnBackend.__init__)."""
    return backend_class(name)(model_runner)


def add_backend_argument(parser, default: str = DEFAULT_BACKEND) -> None:
    """`--attention-backend {mi355,torch_native,fa3}` on an argparse parser (ServerArgs.add_cli_args, utils/args.py:60-141)."""
    parser.add_argument("--attention-backend", type=canonical_backend_name, default=default, metavar="{" + ",".join(BACKEND_CHOICES) + "}",
                        help="attention backend: mi355 (hand-written HIP kernels for MI355X), torch_native (per-sequence SDPA loop), "
                             "fa3 (sgl_kernel FlashAttention-3, CUDA only)")

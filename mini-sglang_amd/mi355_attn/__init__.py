"""mi355_attn - MI355X-native attention backend, KV-store write and page-table metadata for Mini-SGLang.

Product code.  Needs libmi355attn.so (mini-sglang_amd/lib/, built by `make -C mini-sglang_amd/csrc`); there
is no CPU or torch fallback for any device operation.
"""
from .attention import Attention
from .backend import MI355AttnBackend
from .batch import Batch, Mode, Req, SamplingParams
from .kvcache import KVCache
from .page_manager import PageManager
from .radix_cache import PagedRadixCache
from .select import BACKEND_CHOICES, add_backend_argument, backend_class, select_backend
from .tp import HeadShard, all_reduce_rows, shard_heads

__all__ = ["Attention", "MI355AttnBackend", "Batch", "Mode", "Req", "SamplingParams", "KVCache", "PageManager",
           "PagedRadixCache", "HeadShard", "shard_heads", "all_reduce_rows", "select_backend", "backend_class",
           "add_backend_argument", "BACKEND_CHOICES"]

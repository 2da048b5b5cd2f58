"""Harness around the attention path: a synthetic-weight Llama shell and the bench_one_batch counterpart
(SURVEY.md §8 a15).  Everything outside attention / RoPE / KV store / metadata is plain torch (hipBLASLt GEMMs,
F.rms_norm, F.silu): it exists to drive the path the way the reference's model does, not as a product."""

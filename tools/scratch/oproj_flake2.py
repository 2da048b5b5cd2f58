import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from mi355_attn.tp import FusedRowParallel
MODE = os.environ.get("FLAKE_MODE", "none")     # none | sync | l2flush
CASES = [(2, 64, 4096, 2048), (8, 64, 1024, 512)]
bad = 0
for world, T, N, K in CASES:
    f = FusedRowParallel(N, t_cap=256, local_world=world, device="cuda:0", device_epoch=False)
    for call in range(3):
        t = T if call < 2 else max(1, T // 2)
        g = torch.Generator(device="cuda").manual_seed(call)
        xs = [torch.randn((t, K), generator=g, device="cuda").to(torch.bfloat16) for _ in range(world)]
        ws = [(torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(world)]
        want = sum(torch.nn.functional.linear(x.float(), w.float()).to(torch.bfloat16).float() for x, w in zip(xs, ws))
        if MODE == "sync":
            torch.cuda.synchronize()
        if MODE == "l2flush":
            junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); junk.fill_(1); del junk
        outs = f(xs, ws)
        torch.cuda.synchronize()
        nbad = int(((outs[0].float() - want).abs() > 0.05).sum())
        if nbad:
            bad += 1
            m = (outs[0].float() - want).abs() > 0.05
            nz = m.nonzero().tolist()
            rows = sorted({r for r, c in nz}); cols = sorted({c for r, c in nz})
            percol = {c: sum(1 for r2, c2 in nz if c2 == c) for c in cols[:40]}
            print(f"case {(world, T, N, K)} call {call}: {nbad} bad elements; {len(rows)} rows, {len(cols)} cols; per col {percol}", flush=True)
            print("   ptrs ws", [hex(w.data_ptr()) for w in ws], "xs", [hex(x.data_ptr()) for x in xs], "out", hex(outs[0].data_ptr()), flush=True)
    f.close()
print("mode", MODE, "bad calls", bad, flush=True)

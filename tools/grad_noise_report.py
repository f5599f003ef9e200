import sys, torch
sys.path.insert(0, '/root/repo')
from tests import gradcheck, cases
from tests.test_sashimi_training_gpu import TRAIN_CASES, _engine_and_oracle
gpu = torch.device('cuda:0')
for name in ("d32", "snet", "d128"):
    cfg, B = TRAIN_CASES[name]
    net, got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, gpu, 15, 19, 23)
    e_o = gradcheck.errors(got, o32)   # scaled by o32 -- close enough
    e_t = gradcheck.errors(got, truth)
    n_t = gradcheck.errors(o32, truth)
    rows = [(k, e_o[k], e_t[k], n_t[k], kink[k]) for k in got if max(e_o[k], e_t[k]) >= 5e-4]
    print(name, "tensors with err >= 5e-4:", len(rows), "of", len(got))
    for r in sorted(rows, key=lambda r: -r[2]):
        print("   %-50s vs_o32 %.2e  vs_f64 %.2e  o32_vs_f64 %.2e  kink %.2e" % r)

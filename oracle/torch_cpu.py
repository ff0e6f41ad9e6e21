"""torch-cpu restatement used as the timed CPU baseline (bench.py cpu_baseline
leg only; TEST INFRASTRUCTURE, never imported by the product).

It has the STRUCTURE of what the reference runs on the host:
  * six sequential ImprovedProbe-shaped module calls + the softmax / sum /
    threshold of exp_rag.py:406-415 (utils.py:45-57, 389-390);
  * a flat scan as faiss-cpu / torch-cpu would do it: one sgemm for the
    cross term + norms, then top-k (utils.py:378-380 -> IndexFlatL2.search).
faiss itself is not installable in this image."""
import torch
import torch.nn as nn


class ProbeCPU(nn.Module):
    # same layer order as utils.py:45-57 (Linear -> SiLU -> LayerNorm), eval mode
    def __init__(self, d, c=2, h=512):
        super().__init__()
        self.layer_norm_input = nn.LayerNorm(d)
        self.fc1, self.fc2, self.fc3 = nn.Linear(d, h), nn.Linear(h, h), nn.Linear(h, c)
        self.silu = nn.SiLU()
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(h), nn.LayerNorm(h)

    def forward(self, x):
        x = self.layer_norm_input(x)
        x = self.layer_norm1(self.silu(self.fc1(x)))
        x = self.layer_norm2(self.silu(self.fc2(x)))
        return self.fc3(x)


def make_probers(states, d):
    out = []
    for st in states:
        m = ProbeCPU(d)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        out.append(m.eval())
    return out


@torch.no_grad()
def gate(probers, x, ablation=0, theta=0.0):
    """x [L,B,d] float32 CPU -> (probsum [B,2], decision [B])."""
    softmax_f = torch.nn.Softmax(dim=1)
    logits = [p(x[l]) for l, p in enumerate(probers)]
    s = torch.zeros_like(logits[0])
    for n in range(ablation, len(logits)):
        s += softmax_f(logits[n])
    return s, (~(s[:, 0] + theta < s[:, 1])).to(torch.int32)


@torch.no_grad()
def flat_search(xs, xnorm, q, k, metric_l2=True):
    """xs [N,d] float32, xnorm [N] (||x||^2), q [B,d] -> (D [B,k], I [B,k])."""
    dots = q @ xs.T
    if metric_l2:
        sc = (q * q).sum(1, keepdim=True) - 2.0 * dots + xnorm[None, :]
        D, I = torch.topk(sc, k, dim=1, largest=False, sorted=True)
    else:
        D, I = torch.topk(dots, k, dim=1, largest=True, sorted=True)
    return D, I

"""ORACLE — test infrastructure only. CPU restatement of the CRB acquisition arithmetic with the very library calls the
reference makes (sklearn KernelDensity, scipy.stats.entropy / uniform, torch Categorical):
  stage 1 label entropy      pcdet/query_strategies/crb_sampling.py:86-94
  stage 3 prior + greedy     pcdet/query_strategies/crb_sampling.py:247-331
Parity unpinned by reference tests (it has none); the formulas are closed form and the libraries are the reference's."""
import numpy as np
import scipy.stats
import torch
from scipy.stats import uniform
from sklearn.neighbors import KernelDensity
from torch.distributions import Categorical


def label_entropy(pred_labels, num_class):
    """pred_labels: 1-D LongTensor of 1-based class ids -> float"""
    value, counts = torch.unique(pred_labels, return_counts=True)
    if len(value) == 0:
        return 0.0
    p = torch.ones(num_class)
    p[value - 1] = counts.float()
    return float(Categorical(probs=p / sum(counts)).entropy())


def build_prior(density_all, label_all, num_class, alpha=0.95):
    """-> x_axis [C](400,), uniform pdf [C](400,)  (crb_sampling.py:250-260)"""
    unique_labels, label_counts = torch.unique(label_all, return_counts=True)
    sorted_density = [torch.sort(density_all[label_all == u])[0] for u in unique_labels]
    gmax = [int(sorted_density[u][-1]) for u in range(len(unique_labels))]
    ghigh = [int(sorted_density[u][int(alpha * label_counts[u])]) for u in range(len(unique_labels))]
    glow = [int(sorted_density[u][-int(alpha * label_counts[u])]) for u in range(len(unique_labels))]
    x_axis = [np.linspace(-50, int(gmax[i]) + 50, 400) for i in range(num_class)]
    prior = [uniform.pdf(x_axis[i], glow[i], ghigh[i] - glow[i]) for i in range(num_class)]
    return x_axis, prior


def density_greedy(density_list, label_list, x_axis, prior, num_class, select_nums, bandwidth=5):
    """density_list/label_list: per-candidate 1-D tensors -> picked candidate indices (crb_sampling.py:264-331)"""
    density_list = [d.clone() for d in density_list]
    label_list = [l.clone() for l in label_list]
    ids = list(range(len(density_list)))
    picked = []
    sel_d = torch.tensor([])
    sel_l = torch.tensor([])
    scores = []
    for j in range(min(select_nums, len(ids) + len(picked))):
        if j == 0:
            picked.append(ids[0])
            sel_d = torch.cat((sel_d, density_list[0]))
            sel_l = torch.cat((sel_l, label_list[0].float()))
            del density_list[0], label_list[0], ids[0]
            scores.append(-1.0)
            continue
        if not ids:
            break
        best, best_i = -1, None
        for i in range(len(density_list)):
            props = np.zeros(num_class)
            for c in range(num_class):
                if (label_list[i] == c + 1).sum() == 0:
                    props[c] = 1
                else:
                    d = torch.cat((sel_d[sel_l == (c + 1)], density_list[i][label_list[i] == (c + 1)]))
                    kde = KernelDensity(kernel='gaussian', bandwidth=bandwidth).fit(d.cpu().numpy()[:, None])
                    logprob = kde.score_samples(x_axis[c][:, None])
                    kl = scipy.stats.entropy(prior[c], np.exp(logprob))
                    props[c] = 2 / np.pi * np.arctan(np.pi / 2 * kl)
            inv = np.mean(1 - props)
            if inv > best:
                best, best_i = inv, i
        if best_i is None:
            best_i = 0
        sel_d = torch.cat((sel_d, density_list[best_i]))
        sel_l = torch.cat((sel_l, label_list[best_i].float()))
        picked.append(ids[best_i])
        scores.append(best)
        del density_list[best_i], label_list[best_i], ids[best_i]
    return picked, scores


def gt_point_statistics(points_xyz, gt_boxes, num_class):
    """per-class GT point statistics of ONE frame, following detector3d_template.py:236-268 step by step: per class the
    first-hit points-in-boxes over that class's boxes only, `(idx == i).sum()` for every unique index, the first entry of
    the sorted unique list dropped (`[1:]`, meant to skip the -1 bin), then torch.mean / median / var(unbiased=False) with
    NaN -> 0. points_xyz (n,3), gt_boxes (G,8) -> list over classes of (num_bbox, n_counted, mean, median, variance)"""
    import oracle
    out = []
    lab = gt_boxes[:, -1]
    for c in range(num_class):
        m = lab == (c + 1)
        n_cls = int(m.sum())
        if n_cls == 0:
            out.append((0, 0, 0.0, 0.0, 0.0))
            continue
        idx = oracle.points_in_boxes(points_xyz[None].astype(np.float32), gt_boxes[m][None, :, :7].astype(np.float32))[0]
        idx = torch.from_numpy(idx).long()
        cnt = torch.tensor([(idx == i).sum() for i in torch.unique(idx)])[1:].float()
        if cnt.numel() == 0:
            out.append((n_cls, 0, 0.0, 0.0, 0.0))
        else:
            out.append((n_cls, int(cnt.numel()), float(torch.mean(cnt)), float(torch.median(cnt)),
                        float(torch.var(cnt, unbiased=False))))
    return out

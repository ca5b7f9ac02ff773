"""BadgeSampling (pcdet/query_strategies/badge_sampling.py:18-203): hypothetical RPN labels = arg-max of the dense head's
class scores (eval pass, dropout on), then per frame one bs=1 training-mode pass whose RPN classification loss against
those labels gives the gradient of dense_head.conv_cls.weight (18 x 512) as embedding; sklearn kmeans_plusplus
(random_state=0) picks SELECT_NUMS frames.

Differences in mechanics, not in values: labels stay on the device as uint8 (the reference keeps (pool, 211200) int64 on
the host); the gradient is taken with autograd.grad on conv_cls.weight alone (the reference back-propagates the whole
detector and discards everything else); frames are sharded over ranks and the embeddings all-gathered."""
import numpy as np
import torch

from . import scoring
from .pool_eval import PoolEvalStrategy


class BadgeSampling(PoolEvalStrategy):
    MC_DROPOUT = True

    @staticmethod
    def hypothetical_labels(rpn_preds, num_class):
        """(B, H, W, A*num_class) class scores of the dense head -> (B, H*W*A) arg-max labels (badge_sampling.py:86-89)"""
        B = rpn_preds.shape[0]
        return torch.argmax(rpn_preds.reshape(B, -1, num_class), -1)

    @staticmethod
    def head_embedding(dense_head, rpn_preds, labels):
        """one frame (bs = 1, train-mode rpn_preds): gradient of the RPN classification loss against the hypothetical
        labels w.r.t. conv_cls.weight, flattened (badge_sampling.py:146-160; pinned by tests/golden/ref_badge.npz)"""
        new_data = {'box_cls_labels': labels.long().reshape(1, -1), 'cls_preds': rpn_preds}
        loss = dense_head.get_cls_layer_loss(new_data=new_data)[0]
        g, = torch.autograd.grad(loss, dense_head.conv_cls.weight)
        return g.detach().reshape(-1)

    def _rpn_labels(self, batch, pred_dicts, b):
        rpn = pred_dicts[0]['rpn_preds']                                   # (B, H, W, A*num_class) of the whole batch
        return self.hypothetical_labels(rpn, self.detector.dense_head.num_class)[b]

    def grad_embeddings(self, frame_indices, labels):
        model = self.detector
        model.train()
        w = model.dense_head.conv_cls.weight
        out = []
        for k, (chunk, batch) in enumerate(self._batches(self.unlabelled_set, frame_indices, 1)):
            ret, _, _ = model(batch)
            out.append(self.head_embedding(model.dense_head, ret['rpn_preds'], labels[k]))
        model.eval()
        return torch.stack(out, 0) if out else torch.zeros((0, w.numel()), device=w.device)

    def query(self, leave_pbar=True, cur_epoch=None):
        from sklearn.cluster import kmeans_plusplus
        rank, world = self._world()
        n = len(self.pairs)
        mine, _ = scoring.shard_indices(n, rank, world)
        labels = self.eval_pool(self.unlabelled_set, mine, self.unlabelled_loader.batch_size or 1, self._rpn_labels)
        local = self.grad_embeddings(mine, labels)
        emb = self.gather_pool(local, n)
        self.last_embeddings = emb
        _, sel = kmeans_plusplus(emb.cpu().numpy(), n_clusters=self.cfg.ACTIVE_TRAIN.SELECT_NUMS, random_state=0)
        ids = self.unlabelled_set.sample_id_list if hasattr(self.unlabelled_set, 'sample_id_list') else \
            [p[0] for p in self.pairs]
        return [ids[int(i)] for i in np.asarray(sel)]

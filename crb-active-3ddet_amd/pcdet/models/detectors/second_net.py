"""SECONDNet (pcdet/models/detectors/second_net.py:4-34): one-stage detector, the loss is the dense head's.
forward() is Detector3DTemplate.forward."""
from .detector3d_template import Detector3DTemplate


class SECONDNet(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        return loss_rpn, dict(tb_dict, loss_rpn=loss_rpn.detach()), {}

"""PVRCNN (pcdet/models/detectors/pv_rcnn.py:4-43): two-stage detector = RPN (dense head) + keypoint head + RoI head.
forward() is Detector3DTemplate.forward; this class only says how the three losses combine and which second-stage tensors the
training-mode return dict carries (the active-learning code reads them)."""
from .detector3d_template import Detector3DTemplate


class PVRCNN(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def training_outputs(self, batch_dict):
        head = self.roi_head.forward_ret_dict
        return {'rcnn_reg_gt': head['rcnn_reg_gt'], 'rcnn_cls_gt': head['rcnn_cls_labels'],
                'rcnn_cls': batch_dict['rcnn_cls'], 'rcnn_reg': batch_dict['rcnn_reg'],
                'rpn_preds': batch_dict['rpn_preds']}

    def get_training_loss(self):
        total, tb_dict = self.dense_head.get_loss()
        for head in (self.point_head, self.roi_head):
            part, tb_dict = head.get_loss(tb_dict)
            total = total + part
        return total, tb_dict, {}

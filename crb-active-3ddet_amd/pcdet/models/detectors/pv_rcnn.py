"""PVRCNN (pcdet/models/detectors/pv_rcnn.py:4-43)."""
from .detector3d_template import Detector3DTemplate


class PVRCNN(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        if getattr(self, 'pfe', None) is not None and hasattr(self.pfe, 'prefetch_keypoints'):
            self.pfe.prefetch_keypoints(batch_dict)          # FPS on a side stream, joined inside the PFE
        for cur_module in self.module_list:
            batch_dict = cur_module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            ret_dict = {
                'loss': loss,
                'rcnn_reg_gt': self.roi_head.forward_ret_dict['rcnn_reg_gt'],
                'rcnn_cls_gt': self.roi_head.forward_ret_dict['rcnn_cls_labels'],
                'rcnn_cls': batch_dict['rcnn_cls'],
                'rcnn_reg': batch_dict['rcnn_reg'],
                'rpn_preds': batch_dict['rpn_preds'],
            }
            return ret_dict, tb_dict, disp_dict
        pred_dicts, recall_dicts = self.post_processing(batch_dict)
        return pred_dicts, recall_dicts

    def get_training_loss(self):
        disp_dict = {}
        loss_rpn, tb_dict = self.dense_head.get_loss()
        loss_point, tb_dict = self.point_head.get_loss(tb_dict)
        loss_rcnn, tb_dict = self.roi_head.get_loss(tb_dict)
        return loss_rpn + loss_point + loss_rcnn, tb_dict, disp_dict

"""Metric base class and registry, mirroring pycvvdp/vq_metric.py:11-86."""


class vq_exception(Exception):
    def __init__(self, message):
        super().__init__(message)


class vq_metric:
    def predict(self, test_cont, reference_cont, dim_order="BCFHW", frames_per_second=0):
        raise NotImplementedError

    def predict_video_source(self, vid_source):
        raise NotImplementedError

    def quality_unit(self):
        raise NotImplementedError

    def get_info_string(self):
        return None

    def set_base_fname(self, base_fname):
        self.base_fname = base_fname

    def train(self, do_training=True):
        pass

    def short_name(self):
        return self.__class__.__name__.replace("_", "-")

    def export_distogram(self, stats, fname, jod_max=None, base_size=6):
        raise vq_exception(f"Metric {self.short_name()} cannot generate distograms")


vq_metric_dict = dict()


def register_metric(metric_class):
    vq_metric_dict[metric_class.__name__] = metric_class

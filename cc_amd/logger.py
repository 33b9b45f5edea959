"""The part of the reference's logger.py the training / validation loops compute with: AverageMeter (logger.py:62-87).
TermLogger / Writer are terminal UI (blessings, progressbar) and are out of scope ("out of scope")."""


class AverageMeter(object):
    """Running value / sum / count-weighted average of `i` quantities (logger.py:62-87).  Values may be Python numbers or
    0-dim device tensors (the asynchronous validation loops of cc_amd/validate.py keep them on the device until the end)."""

    def __init__(self, i=1, precision=3):
        self.meters = i
        self.precision = precision
        self.reset(self.meters)

    def reset(self, i):
        self.val = [0] * i
        self.avg = [0] * i
        self.sum = [0] * i
        self.count = 0

    def update(self, val, n=1):
        if not isinstance(val, list):
            val = [val]
        assert len(val) == self.meters
        self.count += n
        for k, v in enumerate(val):
            self.val[k] = v
            self.sum[k] = self.sum[k] + v * n
            self.avg[k] = self.sum[k] / self.count

    def __repr__(self):
        val = ' '.join('{:.{}f}'.format(float(v), self.precision) for v in self.val)
        avg = ' '.join('{:.{}f}'.format(float(a), self.precision) for a in self.avg)
        return '{} ({})'.format(val, avg)

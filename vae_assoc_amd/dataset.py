"""Data containers with the reference's batching semantics (/root/reference/dataset.py).

``DataSet.next_batch`` (:22-43): sequential slices; when a slice would run past the end the
data are reshuffled with ``np.random.shuffle`` and the epoch restarts at 0 -- the tail of the
old order is discarded.  ``construct_datasets`` (:45-72): optional shuffle, then split at
int((1-val-test)*N) and int((1-test)*N).  The reference's labelled path is broken (it never
sets ``_num_examples`` when labels are given, :8-15, and slices labels wrongly at :65); here
labels simply work.

``DeviceDataSet`` keeps the sample matrix resident in HBM with the SAME batch order (the
permutation still comes from ``np.random.shuffle`` on the host; only the gather runs on the GPU),
so ``train()`` feeds the step by pointer + row stride without a host->device copy per step.
``extract_images`` / ``extract_jnt_fa_parms`` mirror the reference's pickle-schema readers
(/root/reference/utils.py:142-158, 178-195) on an already-loaded ``{char_key: [arrays]}`` dict.
"""
import numpy as np


class DataSets(object):
    """namespace for the three splits: .train, .validation, .test"""


class DataSet(object):
    """Rows (and optional labels) behind a cursor.  The private names ``_data`` and ``_num_examples`` are part of the contract:
    the reference's ``train()`` reads them (vae_assoc.py:510,543)."""

    def __init__(self, data, labels=None):
        rows = int(data.shape[0])
        if labels is not None:
            assert int(labels.shape[0]) == rows, "data.shape: %s labels.shape: %s" % (data.shape, labels.shape)
        self._data, self._labels = data, labels
        self._num_examples = rows
        self._index_in_epoch = 0          # cursor: first row of the NEXT batch
        self._epochs_completed = 0

    def _take(self, order):
        """re-order the rows (and labels) by an index array"""
        self._data = self._data[order]
        if self._labels is not None:
            self._labels = self._labels[order]

    def _wrap(self):
        """A batch would run past the end: the epoch is over.  ONE ``np.random.shuffle`` of ``arange(N)`` -- the RNG call the reference
        makes at this point (dataset.py:30-31), so that seeded runs visit the same rows -- then the cursor returns to row 0; whatever
        was left of the old order is dropped."""
        order = np.arange(self._num_examples)
        np.random.shuffle(order)
        self._take(order)
        self._epochs_completed += 1
        self._index_in_epoch = 0

    def _slice(self, lo, hi):
        return self._data[lo:hi], (None if self._labels is None else self._labels[lo:hi])

    def next_batch(self, batch_size):
        """``(data[lo:hi], labels[lo:hi] or None)`` of the next ``batch_size`` rows (dataset.py:22-43)"""
        if self._index_in_epoch + batch_size > self._num_examples:
            self._wrap()
            assert batch_size <= self._num_examples
        lo = self._index_in_epoch
        self._index_in_epoch = lo + batch_size
        return self._slice(lo, lo + batch_size)

    def next_batches(self, batch_size, max_batches):
        """``n <= max_batches`` successive ``next_batch`` results as ONE contiguous slice of ``n*batch_size``
        rows: the first call may reshuffle exactly as ``next_batch`` does, the run then extends as far as
        the slices stay consecutive (up to the next wrap).  Returns ``(data, labels, n)``; the state
        afterwards is what ``n`` ``next_batch`` calls leave behind."""
        self.next_batch(batch_size)
        lo = self._index_in_epoch - batch_size
        n = 1 + max(0, min(int(max_batches) - 1, (self._num_examples - self._index_in_epoch) // batch_size))
        self._index_in_epoch = lo + n * batch_size
        d, l = self._slice(lo, self._index_in_epoch)
        return d, l, n


def construct_datasets(data, labels=None, shuffle=True, validation_ratio=.1, test_ratio=.1):
    """Optional shuffle (one ``np.random.shuffle`` of ``arange(N)``), then train | validation | test at
    ``int((1 - validation_ratio - test_ratio) N)`` and ``int((1 - test_ratio) N)`` (dataset.py:45-72)."""
    rows = int(data.shape[0])
    if shuffle:
        order = np.arange(rows)
        np.random.shuffle(order)
        data = data[order]
        labels = None if labels is None else labels[order]
    cut_val, cut_test = int((1 - validation_ratio - test_ratio) * rows), int((1 - test_ratio) * rows)
    out = DataSets()
    for name, lo, hi in (("train", 0, cut_val), ("validation", cut_val, cut_test), ("test", cut_test, rows)):
        setattr(out, name, DataSet(data[lo:hi, :], None if labels is None else labels[lo:hi]))
    return out


class DeviceDataSet(DataSet):
    """DataSet whose ``_data`` is a torch tensor on the GPU; ``next_batch`` returns device views.  The permutation still comes from
    ``np.random.shuffle`` on the host (same stream as the reference); only the gather runs on the device."""

    def __init__(self, data, labels=None, device=None):
        import torch
        if not torch.is_tensor(data):
            data = torch.as_tensor(np.ascontiguousarray(data, dtype=np.float32))
        data = data.to(device if device is not None else "cuda", dtype=torch.float32)
        DataSet.__init__(self, data, labels)

    def _take(self, order):
        import torch
        self._data = self._data[torch.as_tensor(order, device=self._data.device)]
        if self._labels is not None:
            self._labels = self._labels[order]


def to_device(data_sets, device=None):
    """Move the three splits of ``construct_datasets`` into HBM (same order, same batching)."""
    out = DataSets()
    for name in ("train", "validation", "test"):
        ds = getattr(data_sets, name)
        setattr(out, name, DeviceDataSet(ds._data, ds._labels, device))
    return out


def extract_images(data, only_digits=True, dtype=np.float32):
    """utils.py:142-158: characters ordered by their LAST key character, images flattened and scaled by 1/255."""
    images = []
    for char in sorted(data.keys(), key=lambda k: k[-1]):
        if only_digits and ord(char[-1]) > 57:
            continue
        images += [np.asarray(d).flatten().astype(dtype) * 1. / 255. for d in data[char]]
    return np.array(images)


def extract_jnt_fa_parms(data, only_digits=True):
    """utils.py:178-195: characters ordered by the WHOLE key (note: not the order extract_images uses --
    pairing the two relies on both orders agreeing, as in the reference); returns (parms, mean, std)."""
    fa_parms = []
    for char in sorted(data.keys()):
        if only_digits and ord(char[-1]) > 57:
            continue
        fa_parms += [d for d in data[char]]
    fa_parms = np.array(fa_parms)
    return fa_parms, np.mean(fa_parms, axis=0), np.std(fa_parms, axis=0)

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
    pass


class DataSet(object):
    def __init__(self, data, labels=None):
        if labels is not None:
            assert data.shape[0] == labels.shape[0], (
                'data.shape: %s labels.shape: %s' % (data.shape, labels.shape))
        self._num_examples = data.shape[0]
        self._data = data
        self._labels = labels
        self._epochs_completed = 0
        self._index_in_epoch = 0

    def next_batch(self, batch_size):
        """Return the next `batch_size` examples from this data set."""
        start = self._index_in_epoch
        self._index_in_epoch += batch_size
        if self._index_in_epoch > self._num_examples:
            self._epochs_completed += 1          # finished epoch
            perm = np.arange(self._num_examples)
            np.random.shuffle(perm)              # shuffle the data
            self._data = self._data[perm]
            if self._labels is not None:
                self._labels = self._labels[perm]
            start = 0                            # start next epoch
            self._index_in_epoch = batch_size
            assert batch_size <= self._num_examples
        end = self._index_in_epoch
        if self._labels is not None:
            return self._data[start:end], self._labels[start:end]
        return self._data[start:end], None

    def next_batches(self, batch_size, max_batches):
        """``n <= max_batches`` successive ``next_batch`` results as ONE contiguous slice of ``n*batch_size``
        rows: the first call may reshuffle exactly as ``next_batch`` does, the run then extends as far as
        the slices stay consecutive (up to the next wrap).  Returns ``(data, labels, n)``; the state
        afterwards is what ``n`` ``next_batch`` calls leave behind."""
        self.next_batch(batch_size)
        start = self._index_in_epoch - batch_size
        n = 1 + max(0, min(int(max_batches) - 1, (self._num_examples - self._index_in_epoch) // batch_size))
        self._index_in_epoch = start + n * batch_size
        end = self._index_in_epoch
        return self._data[start:end], (self._labels[start:end] if self._labels is not None else None), n


def construct_datasets(data, labels=None, shuffle=True, validation_ratio=.1, test_ratio=.1):
    data_sets = DataSets()
    if shuffle:
        perm = np.arange(data.shape[0])
        np.random.shuffle(perm)
        data_shuffled = data[perm]
        labels_shuffled = labels[perm] if labels is not None else None
    else:
        data_shuffled = data
        labels_shuffled = labels
    n = data_shuffled.shape[0]
    test_start_idx = int((1 - test_ratio) * n)
    validation_start_idx = int((1 - validation_ratio - test_ratio) * n)
    lab = (lambda a, b: labels_shuffled[a:b]) if labels is not None else (lambda a, b: None)
    data_sets.train = DataSet(data_shuffled[:validation_start_idx, :], lab(0, validation_start_idx))
    data_sets.validation = DataSet(data_shuffled[validation_start_idx:test_start_idx, :],
                                   lab(validation_start_idx, test_start_idx))
    data_sets.test = DataSet(data_shuffled[test_start_idx:, :], lab(test_start_idx, n))
    return data_sets


class DeviceDataSet(DataSet):
    """DataSet whose ``_data`` is a torch tensor on the GPU; ``next_batch`` returns device views."""

    def __init__(self, data, labels=None, device=None):
        import torch
        if not torch.is_tensor(data):
            data = torch.as_tensor(np.ascontiguousarray(data, dtype=np.float32))
        data = data.to(device if device is not None else "cuda", dtype=torch.float32)
        DataSet.__init__(self, data, labels)

    def next_batch(self, batch_size):
        import torch
        start = self._index_in_epoch
        self._index_in_epoch += batch_size
        if self._index_in_epoch > self._num_examples:
            self._epochs_completed += 1
            perm = np.arange(self._num_examples)
            np.random.shuffle(perm)                   # same host RNG stream as the reference (dataset.py:30-31)
            self._data = self._data[torch.as_tensor(perm, device=self._data.device)]
            if self._labels is not None:
                self._labels = self._labels[perm]
            start = 0
            self._index_in_epoch = batch_size
            assert batch_size <= self._num_examples
        end = self._index_in_epoch
        return self._data[start:end], (self._labels[start:end] if self._labels is not None else None)


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

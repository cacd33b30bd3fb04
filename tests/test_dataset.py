"""Host-side data containers against the batch order recorded from the
reference's own dataset.py (tests/golden/make_golden.py, fixture dataset_order.npz)."""
import os

import numpy as np

from conftest import GOLDEN
from vae_assoc_amd import dataset


def test_batch_order_matches_reference_fixture():
    z = np.load(os.path.join(GOLDEN, "dataset_order.npz"), allow_pickle=False)
    np.random.seed(int(z["seed"]))
    ds = dataset.construct_datasets(z["data"].copy(), validation_ratio=.1, test_ratio=.1)
    assert ds.train._data.shape[0] == int(z["n_train"])
    assert np.array_equal(ds.validation._data[:, 0], z["validation"])
    assert np.array_equal(ds.test._data[:, 0], z["test"])
    B = int(z["batch"])
    for want in z["batches"]:
        got, lab = ds.train.next_batch(B)
        assert lab is None and np.array_equal(got[:, 0], want)
    assert ds.train._epochs_completed == 3          # 17 batches of 8 over 42 rows: wraps at 6, 11, 16


def test_tail_is_dropped_and_reshuffled_on_wrap():
    np.random.seed(0)
    ds = dataset.DataSet(np.arange(10)[:, None].astype(float))
    a, _ = ds.next_batch(4)
    b, _ = ds.next_batch(4)
    assert list(a[:, 0]) == [0, 1, 2, 3] and list(b[:, 0]) == [4, 5, 6, 7]
    c, _ = ds.next_batch(4)                        # would need rows 8..11 -> reshuffle, restart at 0
    assert ds._epochs_completed == 1 and ds._index_in_epoch == 4 and len(c) == 4
    assert sorted(ds._data[:, 0]) == list(range(10))


def test_next_batches_is_a_run_of_next_batch_calls():
    """next_batches(B, k) = the next n <= k next_batch(B) results as one slice, same state afterwards."""
    x = np.arange(53, dtype=float)[:, None]
    np.random.seed(7)
    a = dataset.DataSet(x.copy(), labels=x.copy() * 2)
    singles = [a.next_batch(5) for _ in range(40)]
    np.random.seed(7)
    b = dataset.DataSet(x.copy(), labels=x.copy() * 2)
    got, asked = 0, [3, 1, 100, 4, 2, 7, 100, 100, 100]
    while got < 40:
        d, l, n = b.next_batches(5, min(asked[got % len(asked)], 40 - got))
        assert n >= 1 and d.shape[0] == 5 * n
        for j in range(n):
            assert np.array_equal(d[5 * j:5 * j + 5], singles[got + j][0]) and np.array_equal(l[5 * j:5 * j + 5], singles[got + j][1])
        got += n
    assert (a._index_in_epoch, a._epochs_completed) == (b._index_in_epoch, b._epochs_completed)
    assert np.array_equal(a._data, b._data)


def test_labels_travel_with_data():
    np.random.seed(3)
    x = np.arange(20, dtype=float).reshape(10, 2)
    ds = dataset.construct_datasets(x, labels=x[:, :1] * 10)
    d, l = ds.train.next_batch(3)
    assert np.array_equal(l[:, 0], d[:, 0] * 10)


def test_split_points():
    ds = dataset.construct_datasets(np.zeros((25, 2)), shuffle=False)
    assert (ds.train._data.shape[0], ds.validation._data.shape[0], ds.test._data.shape[0]) == (20, 2, 3)


def test_extractors_follow_reference_key_orders():
    """utils.py:153 sorts by the last key character, :186 by the whole key; digits only by default."""
    d = {"b_1": [np.full((2, 2), 10)], "a_2": [np.full((2, 2), 20)], "c_a": [np.full((2, 2), 30)]}
    imgs = dataset.extract_images(d)
    assert imgs.shape == (2, 4) and np.allclose(imgs[:, 0], [10 / 255., 20 / 255.])          # "..1" before "..2"
    fa, mean, std = dataset.extract_jnt_fa_parms({k: [v[0].ravel().astype(float)] for k, v in d.items()})
    assert np.allclose(fa[:, 0], [20, 10]) and np.allclose(mean, 15) and np.allclose(std, 5)   # "a_2" before "b_1"
    assert dataset.extract_images(d, only_digits=False).shape[0] == 3

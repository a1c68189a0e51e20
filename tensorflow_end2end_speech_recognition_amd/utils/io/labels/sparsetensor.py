"""Label wire format at the model boundary -- mirror of
utils/io/labels/sparsetensor.py:12-78 (list2sparsetensor / sparsetensor2list),
without the tensorflow import the reference does at module top."""
import numpy as np


def list2sparsetensor(labels, padded_value):
    """labels [B, max_label_len] padded with `padded_value` ->
    [indices int64 [n,2], values, dense_shape int64 [2]] (sparsetensor.py:12-39)."""
    if padded_value is None:
        dtype_values = np.uint8
    else:
        dtype_values = np.int32
    indices, values = [], []
    for i_utt, each_label in enumerate(labels):
        for i_l, l in enumerate(each_label):
            if l == padded_value:
                break
            indices.append([i_utt, i_l])
            values.append(l)
    dense_shape = [len(labels), np.asarray(indices).max(0)[1] + 1]
    return [np.array(indices, dtype=np.int64), np.array(values, dtype=dtype_values),
            np.array(dense_shape, dtype=np.int64)]


def sparsetensor2list(labels_st, batch_size):
    """sparsetensor.py:42-78.  Unlike the reference (TODO at :66-70) rows that decode to
    nothing are returned as empty arrays instead of shifting the later rows."""
    indices = np.asarray(labels_st[0]).reshape(-1, 2)
    values = np.asarray(labels_st[1])
    if batch_size == 1:
        return values.reshape((1, -1))
    labels = [[] for _ in range(batch_size)]
    for (b, _), v in zip(indices, values):
        labels[int(b)].append(v)
    return [np.asarray(l, dtype=values.dtype) for l in labels]


def sparse_to_flat(labels_st, batch_size):
    """(indices, values, shape) -> (flat int32 values in row order, offsets [B+1], max_len)."""
    indices = np.asarray(labels_st[0]).reshape(-1, 2)
    values = np.asarray(labels_st[1]).astype(np.int32)
    order = np.lexsort((indices[:, 1], indices[:, 0])) if len(indices) else np.zeros(0, np.int64)
    counts = np.bincount(indices[:, 0].astype(np.int64), minlength=batch_size) if len(indices) \
        else np.zeros(batch_size, np.int64)
    offsets = np.zeros(batch_size + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(counts)
    return values[order], offsets, int(counts.max()) if batch_size else 0


def dense_to_flat(labels, padded_value=-1):
    """[B, Lmax] padded dense labels -> (flat, offsets, max_len); stops at the first pad."""
    flat, offsets = [], [0]
    for row in np.asarray(labels):
        n = 0
        for v in row:
            if v == padded_value:
                break
            flat.append(int(v))
            n += 1
        offsets.append(offsets[-1] + n)
    lens = np.diff(offsets)
    return (np.asarray(flat, dtype=np.int32), np.asarray(offsets, dtype=np.int32),
            int(lens.max()) if len(lens) else 0)

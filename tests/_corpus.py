"""Test helper: a generated corpus in the directory layout the TIMIT recipe reads."""
import os

import numpy as np


def make_timit_like(root, rng, n_train=24, n_dev=6, n_test=4, feat=6):
    """A corpus in the directory layout the TIMIT recipe reads (examples/timit/data/load_dataset_ctc.py): every
    phone owns a feature vector held for 2-3 frames."""
    import pickle
    from examples.timit.metrics.mapping_files import phone_tables
    from examples.timit.metrics.mapping import Map2phone39
    p61 = phone_tables()['phone61']
    use = [p61.index(p) for p in ('aa', 'b', 'iy', 'k', 's', 'h#', 'ao', 'tcl', 't')]
    protos = rng.randn(61, feat).astype(np.float32) * 1.5
    to39 = Map2phone39('phone61')
    for data_type, n in (('train', n_train), ('dev', n_dev), ('test', n_test)):
        os.makedirs(os.path.join(root, 'inputs', data_type))
        lt = 'phone39' if data_type == 'test' else 'phone61'
        os.makedirs(os.path.join(root, 'labels', data_type, lt))
        frame_num = {}
        for i in range(n):
            lab = [use[j] for j in rng.randint(0, len(use), size=rng.randint(3, 6))]
            x = np.concatenate([np.repeat(protos[c][None], rng.randint(2, 4), 0) for c in lab], 0)
            x = (x + 0.1 * rng.randn(*x.shape)).astype(np.float32)
            name = '%s_utt%02d' % (data_type, i)
            np.save(os.path.join(root, 'inputs', data_type, name + '.npy'), x)
            if data_type == 'test':       # the test set stores the 39-phone transcript as a string
                np.save(os.path.join(root, 'labels', data_type, lt, name + '.npy'),
                        np.array(' '.join(to39([p61[c] for c in lab]))))
            else:
                np.save(os.path.join(root, 'labels', data_type, lt, name + '.npy'), np.asarray(lab, dtype=np.int32))
            frame_num[name] = x.shape[0]
        with open(os.path.join(root, 'inputs', data_type, 'frame_num.pickle'), 'wb') as f:
            pickle.dump(frame_num, f)

"""Test helper: a generated corpus in the directory layout the TIMIT recipe reads."""
import os

import numpy as np


def make_timit_like(root, rng, n_train=24, n_dev=6, n_test=4, feat=6, multitask=False):
    """A corpus in the directory layout the TIMIT recipes read (examples/timit/data/load_dataset_*.py): every
    phone owns a feature vector held for 2-3 frames.  labels/<set>/phone61 (index arrays; the test set is scored on
    labels/test/phone39, stored as a string) and labels/<set>/character (one letter per phone, '_' for silence; the
    test transcript is a string).  multitask=True stores the test set's phone39 labels as index arrays, which is
    what the multitask recipe's sub task reads."""
    import pickle
    from examples.timit.metrics.mapping_files import phone_tables, character_tables
    from examples.timit.metrics.mapping import Map2phone39
    tables = phone_tables()
    p61, p39 = tables['phone61'], tables['phone39']
    chars = character_tables()['character']
    letter = {'aa': 'a', 'b': 'b', 'iy': 'i', 'k': 'k', 's': 's', 'h#': '_', 'ao': 'o', 'tcl': '', 't': 't'}
    use = [p61.index(p) for p in letter]
    protos = rng.randn(61, feat).astype(np.float32) * 1.5
    to39 = Map2phone39('phone61')
    for data_type, n in (('train', n_train), ('dev', n_dev), ('test', n_test)):
        os.makedirs(os.path.join(root, 'inputs', data_type))
        lt = 'phone39' if data_type == 'test' else 'phone61'
        os.makedirs(os.path.join(root, 'labels', data_type, lt))
        os.makedirs(os.path.join(root, 'labels', data_type, 'character'))
        frame_num = {}
        for i in range(n):
            lab = [use[j] for j in rng.randint(0, len(use), size=rng.randint(3, 6))]
            x = np.concatenate([np.repeat(protos[c][None], rng.randint(2, 4), 0) for c in lab], 0)
            x = (x + 0.1 * rng.randn(*x.shape)).astype(np.float32)
            name = '%s_utt%02d' % (data_type, i)
            np.save(os.path.join(root, 'inputs', data_type, name + '.npy'), x)
            text = ''.join(letter[p61[c]] for c in lab).strip('_') or 'a'
            if data_type == 'test':       # the test set stores transcripts as strings
                ph39 = to39([p61[c] for c in lab])
                np.save(os.path.join(root, 'labels', data_type, lt, name + '.npy'),
                        np.asarray([p39.index(p) for p in ph39], dtype=np.int32) if multitask
                        else np.array(' '.join(ph39)))
                np.save(os.path.join(root, 'labels', data_type, 'character', name + '.npy'), np.array(text))
            else:
                np.save(os.path.join(root, 'labels', data_type, lt, name + '.npy'), np.asarray(lab, dtype=np.int32))
                np.save(os.path.join(root, 'labels', data_type, 'character', name + '.npy'),
                        np.asarray([chars.index(c) for c in text], dtype=np.int32))
            frame_num[name] = x.shape[0]
        with open(os.path.join(root, 'inputs', data_type, 'frame_num.pickle'), 'wb') as f:
            pickle.dump(frame_num, f)


def make_librispeech_like(root, rng, n_train=12, n_other=4, feat=6, size='train100h'):
    """<root>/inputs/<size>/<data_type>/{frame_num.pickle, <speaker>/<utt>.npy} and the matching labels tree
    (character indices; the two test sets store the transcript as a string), as the Librispeech recipe reads them.
    Every character owns a feature vector held for 2 frames."""
    import pickle
    from examples.timit.metrics.mapping_files import character_tables
    chars = character_tables()['character']
    protos = rng.randn(len(chars), feat).astype(np.float32) * 1.5
    words = ['the', 'cat', 'sat', 'on', 'a', 'mat', 'dog', 'ran']
    for data_type, n in (('train', n_train), ('dev_clean', n_other), ('dev_other', n_other), ('test_clean', n_other),
                         ('test_other', n_other)):
        inp = os.path.join(root, 'inputs', size, data_type)
        lab = os.path.join(root, 'labels', size, data_type, 'character')
        frame_num = {}
        for i in range(n):
            speaker = str(100 + i % 3)
            os.makedirs(os.path.join(inp, speaker), exist_ok=True)
            os.makedirs(os.path.join(lab, speaker), exist_ok=True)
            text = '_'.join(words[j] for j in rng.randint(0, len(words), size=rng.randint(1, 4)))
            idx = [chars.index(c) for c in text]
            x = np.concatenate([np.repeat(protos[c][None], 2, 0) for c in idx], 0)
            x = (x + 0.1 * rng.randn(*x.shape)).astype(np.float32)
            name = '%s-%d-%04d' % (speaker, 7, i)
            np.save(os.path.join(inp, speaker, name + '.npy'), x)
            np.save(os.path.join(lab, speaker, name + '.npy'),
                    np.array(text) if 'test' in data_type else np.asarray(idx, dtype=np.int32))
            # word-level targets of the same transcript (label type word_freq10; index = position in `words`)
            wdir = os.path.join(root, 'labels', size, data_type, 'word_freq10', speaker)
            os.makedirs(wdir, exist_ok=True)
            widx = [words.index(w) for w in text.split('_')]
            np.save(os.path.join(wdir, name + '.npy'),
                    np.array(text) if 'test' in data_type else np.asarray(widx, dtype=np.int32))
            frame_num[name] = x.shape[0]
        with open(os.path.join(inp, 'frame_num.pickle'), 'wb') as f:
            pickle.dump(frame_num, f)

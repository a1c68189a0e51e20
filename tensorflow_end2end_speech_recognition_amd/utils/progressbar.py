"""Optional tqdm wrapping of iterators / generators (utils/progressbar.py:11-23 of the reference)."""


def wrap_iterator(iterator, progressbar):
    if progressbar:
        from tqdm import tqdm
        return tqdm(iterator)
    return iterator


def wrap_generator(generator, progressbar, total):
    if progressbar:
        from tqdm import tqdm
        return tqdm(generator, total=total)
    return generator

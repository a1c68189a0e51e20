"""Iterator bookkeeping shared by every dataset class: epoch / iteration counters, the pool of utterances not yet
drawn in the current epoch (`rest`), and the token map that supplies <SOS> / <EOS> for the attention models.
Same attributes and methods as the reference's utils/dataset/base.py:11-61 Base (the batch streams built on top of
it are pinned to the reference's by tests/golden/datasets_v1.npz)."""


def _read_token_map(path):
    """`<token>  <index>` per line -> {token: index}."""
    table = {}
    with open(path, 'r') as f:
        for row in f:
            fields = row.split()
            if len(fields) >= 2:
                table[fields[0]] = int(fields[1])
    return table


class Base(object):

    def __init__(self, *args, **kwargs):
        self.epoch, self.iteration, self.is_new_epoch = 0, 0, False
        path = kwargs.get('map_file_path')
        self.map_dict = _read_token_map(path) if path is not None else {}

    # ---- size / access
    def __len__(self):
        return len(self.input_paths)

    def __getitem__(self, index):
        return (self.input_list[index], self.label_list[index])

    @property
    def epoch_detail(self):
        """Fraction of epochs consumed so far (iteration counts utterances)."""
        return self.iteration / len(self)

    @property
    def sos_index(self):
        return self.map_dict['<']

    @property
    def eos_index(self):
        return self.map_dict['>']

    # ---- iteration
    def reset(self):
        """Start a new pass: every utterance is available again."""
        self.rest = set(range(len(self)))

    def __iter__(self):
        return self

    def __next__(self):
        raise NotImplementedError

    def next(self, batch_size=None):
        return self.__next__(batch_size)

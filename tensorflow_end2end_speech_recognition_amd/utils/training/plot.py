"""Loss / label-error-rate curves of a run -- the calls of utils/training/plot.py:20-78 (plot_loss, plot_ler).
The numbers always go to `loss.csv` / `ler.csv` in `save_path`; the `.png` next to them is drawn only when
matplotlib is importable (it is not part of this image)."""
import os


def _write_csv(path, header, steps, train, dev):
    with open(path, 'w') as f:
        f.write(header + '\n')
        for s, a, b in zip(steps, train, dev):
            f.write('%d,%f,%f\n' % (int(s), float(a), float(b)))


def _draw(path, steps, train, dev, ylabel, title):
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except ImportError:
        return False
    plt.figure()
    plt.plot(steps, train, label='Train')
    plt.plot(steps, dev, label='Dev')
    plt.xlabel('step')
    plt.ylabel(ylabel)
    plt.title(title)
    plt.legend(loc='upper right')
    plt.savefig(path, dpi=150)
    plt.close()
    return True


def plot_loss(train_losses, dev_losses, steps, save_path):
    _write_csv(os.path.join(save_path, 'loss.csv'), 'step,train,dev', steps, train_losses, dev_losses)
    return _draw(os.path.join(save_path, 'loss.png'), steps, train_losses, dev_losses, 'loss', 'Loss')


def plot_ler(train_lers, dev_lers, steps, label_type, save_path):
    _write_csv(os.path.join(save_path, 'ler.csv'), 'step,train,dev', steps, train_lers, dev_lers)
    name = 'CER' if 'char' in label_type else ('WER' if 'word' in label_type else 'PER')
    return _draw(os.path.join(save_path, 'ler.png'), steps, train_lers, dev_lers, name, name)

"""@measure_time: print the wall-clock time of a call (utils/measure_time_func.py:12-19 of the reference)."""
import functools
import time


def measure_time(func):
    @functools.wraps(func)
    def timed(*args, **kwargs):
        start = time.time()
        out = func(*args, **kwargs)
        print('Takes %.3f sec' % (time.time() - start))
        return out
    return timed

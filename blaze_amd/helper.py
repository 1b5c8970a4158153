"""Host helpers on the step-3 path, same names and behaviour as the reference's
/root/reference/blaze/helper.py (reverse_complement :14-25, batch_iterator :275-289,
fastq_parser :292-301, err_msg/green_msg :27-49).  The ordered process pool
(multiprocessing_submit, :111-205) has no counterpart: device batches replace it."""
from collections import namedtuple

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def reverse_complement(seq):
    """Complement A/C/G/T (either case), leave every other character, reverse."""
    return seq.translate(_COMP)[::-1]


def err_msg(msg, printit=False):
    text = '\033[91m' + msg + '\033[0m'
    if printit:
        print(text)
    else:
        return text


def warning_msg(msg, printit=False):
    text = '\033[93m' + msg + '\033[0m'
    if printit:
        print(text)
    else:
        return text


def green_msg(msg, printit=False):
    text = '\033[92m' + msg + '\033[0m'
    if printit:
        print(text)
    else:
        return text


def batch_iterator(iterator, batch_size):
    """Lists of up to batch_size consecutive items; the last one may be shorter."""
    batch = []
    for entry in iterator:
        batch.append(entry)
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


read_tuple = namedtuple('read_tuple', ['id', 'seq', 'q_letter'])


def fastq_parser(file_handle):
    """4-line FASTQ records -> (id = first token after '@', seq, quality), stripped."""
    while True:
        title = next(file_handle, None)
        if title is None:
            break
        seq = next(file_handle)
        next(file_handle)          # '+'
        qual = next(file_handle)
        yield read_tuple(title[1:].split()[0], seq.strip(), qual.strip())


def gzip_fastq_member(text: bytes) -> bytes:
    """One gzip member of `text` by the native FASTQ-aware deflate encoder (what the native host writes per batch;
    replaces gzip.compress(out_buffer.encode()), read_assignment.py:244-245: same decompressed bytes).  Host only."""
    import ctypes
    from . import _lib
    L = _lib.load()
    cap = L.bz_gzip_fastq_bound(len(text))
    out = ctypes.create_string_buffer(int(cap))
    n = ctypes.c_int64(0)
    _lib.check(L.bz_gzip_fastq_member(text, len(text), out, cap, ctypes.byref(n)))
    return out.raw[:n.value]

"""Mirror of the reference's data/dataset_utils.py helpers that sit on the path."""
import os
import struct
import threading

import numpy as np


def scalar_amplitude_to_dB(x):
    """data/dataset_utils.py:39-43: 20 * log10(x)."""
    return 20 * np.log10(x)


def scalar_dB_to_amplitude(x):
    """data/dataset_utils.py:46-50: 10 ** (0.5 * x)  (sic -- not x/20; kept for parity with trained models)."""
    return np.power(10.0, 0.5 * x)


def split_songlist(songlist, train_val_test_split: tuple = (0.8, 0.2, 0.0), summary: bool = True) -> tuple:
    """data/dataset_utils.py:6-36: random train / val / test split of a song list (numpy's global RNG, as the reference:
    seed it with np.random.seed for a repeatable split).  Train and val are drawn without replacement in that order,
    test is whatever is left."""
    assert sum(train_val_test_split) == 1, 'train/val/test split should sum to 1'
    sizes = [round(len(songlist) * share) for share in train_val_test_split]
    pool, parts = set(songlist), []
    for size in sizes[:2]:
        drawn = list(np.random.choice(list(pool), size, replace=False))
        pool = pool.difference(drawn)
        parts.append(drawn)
    parts.append(list(pool))
    if summary:
        print('Dataset split:')
        print('=' * 80)
        for k, (label, part) in enumerate(zip(('Train', 'Val', 'Test'), parts)):
            if k:
                print('-' * 80)
            print('{}: {} tracks'.format(label, sizes[k]))
            print(part)
    return tuple(parts)


# ---- WAV decoding (stands in for soundfile.read / librosa.load, which are not in the image) ------------------------
_WAVE_FORMAT_PCM, _WAVE_FORMAT_IEEE_FLOAT, _WAVE_FORMAT_EXTENSIBLE = 1, 3, 0xFFFE
_headers = {}


_fds = {}                      # (path, mtime, size) -> read-only descriptor kept open for positioned reads (first _FDS_MAX files)
_fd_of_path = {}               # path -> the key its cached descriptor was opened under
_FDS_MAX = 256
_fds_lock = threading.Lock()
_fd_users = 0                  # ingest passes (MultitrackAudioDataset.iter_batches) currently reading through the cache
_fd_pins = {}                  # descriptor -> readers that hold it right now (between _get_fd and _put_fd)
_fd_doomed = set()             # descriptors taken out of the cache while pinned: the last reader closes them


def _retire(fd):
    """(lock held) A descriptor leaves the cache: closed now, or by its last reader if someone is inside a pread on it --
    never under a reader (EBADF, or, once the number has been reused, bytes of another file)."""
    if _fd_pins.get(fd):
        _fd_doomed.add(fd)
    else:
        os.close(fd)


def _get_fd(path, h):
    """(descriptor, cached?): cached for the first _FDS_MAX files, a fresh one beyond that.  Every reader -- an ingest pass'
    decode threads, __getitem__ on another thread, a second dataset -- PINS the descriptor it was handed until _put_fd():
    close_cached_files() and the stale-path check retire a pinned descriptor instead of closing it."""
    key = h['_key']
    with _fds_lock:
        fd = _fds.get(key)
        if fd is None:
            stale = _fd_of_path.get(path)
            if stale is not None and stale != key:
                old = _fds.pop(stale, None)            # the path was rewritten: its old descriptor points at a dead inode
                if old is not None:
                    _retire(old)
                _headers.pop(stale, None)
            if len(_fds) >= _FDS_MAX:
                return os.open(path, os.O_RDONLY), False
            fd = _fds[key] = os.open(path, os.O_RDONLY)
            _fd_of_path[path] = key
        _fd_pins[fd] = _fd_pins.get(fd, 0) + 1
        return fd, True


def _put_fd(fd, cached):
    if not cached:
        os.close(fd)
        return
    with _fds_lock:
        n = _fd_pins[fd] - 1
        if n:
            _fd_pins[fd] = n
        else:
            del _fd_pins[fd]
            if fd in _fd_doomed:
                _fd_doomed.discard(fd)
                os.close(fd)


def _after_fork_in_child():
    """A DataLoader worker is a fork taken while decode threads of the parent may sit inside _get_fd: the child gets a fresh
    lock and no pins (the inherited descriptors stay valid: positioned reads share no file offset)."""
    global _fds_lock
    _fds_lock = threading.Lock()
    _fd_pins.clear()
    _fd_doomed.clear()


os.register_at_fork(after_in_child=_after_fork_in_child)


def cached_files_begin():
    """An ingest pass starts reading through the descriptor cache (paired with close_cached_files())."""
    global _fd_users
    with _fds_lock:
        _fd_users += 1


def close_cached_files(force=False):
    """End of an ingest pass: when no other pass is running, the cache is emptied (a long-lived process that walks many
    datasets, or rewrites its files between passes, would otherwise hold up to _FDS_MAX descriptors for good); a descriptor
    some reader outside the passes still holds is closed by that reader.  force=True empties the cache regardless of the
    pass count (tests)."""
    global _fd_users
    with _fds_lock:
        _fd_users = max(0, _fd_users - 1)
        if _fd_users == 0 or force:
            for fd in _fds.values():
                _retire(fd)
            _fds.clear()
            _fd_of_path.clear()


def wav_header(path):
    """Parses the RIFF chunks of a WAV file once: {'tag' (1 integer PCM / 3 IEEE float), 'channels', 'rate', 'bits',
    'frame_bytes', 'data_offset', 'frames'}.  WAVE_FORMAT_EXTENSIBLE files (what DAWs write for 24-bit / multichannel
    audio; the stdlib ``wave`` module rejects them) are resolved through their sub-format."""
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)
    if key in _headers:
        return _headers[key]
    with open(path, 'rb') as fh:
        riff = fh.read(12)
        if len(riff) < 12 or riff[:4] != b'RIFF' or riff[8:12] != b'WAVE':
            raise ValueError('%s is not a RIFF/WAVE file' % path)
        fmt = data = None
        while fmt is None or data is None:
            hdr = fh.read(8)
            if len(hdr) < 8:
                break
            cid, size = hdr[:4], struct.unpack('<I', hdr[4:])[0]
            if cid == b'fmt ':
                fmt = fh.read(size)
                if size & 1:
                    fh.seek(1, 1)
            elif cid == b'data':
                data = (fh.tell(), size)
                fh.seek(size + (size & 1), 1)
            else:
                fh.seek(size + (size & 1), 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError('%s: missing fmt or data chunk' % path)
    tag, channels, rate, _, frame_bytes, bits = struct.unpack('<HHIIHH', fmt[:16])
    if tag == _WAVE_FORMAT_EXTENSIBLE and len(fmt) >= 26:
        tag = struct.unpack('<H', fmt[24:26])[0]          # first two bytes of the sub-format GUID
    if tag not in (_WAVE_FORMAT_PCM, _WAVE_FORMAT_IEEE_FLOAT) or channels < 1 or frame_bytes != channels * ((bits + 7) // 8):
        raise ValueError('%s: unsupported WAV encoding (format tag %d, %d bits)' % (path, tag, bits))
    offset, size = data
    size = min(size, st.st_size - offset)                  # streamed files may carry a placeholder size
    h = dict(tag=tag, channels=channels, rate=rate, bits=bits, frame_bytes=frame_bytes, data_offset=offset,
             frames=size // frame_bytes, _key=key)
    _headers[key] = h
    return h


def _decode(raw, h, dtype):
    """Interleaved sample bytes -> dtype array in [-1, 1) with soundfile's integer normalisation (divide by 2^(bits-1))."""
    tag, width = h['tag'], (h['bits'] + 7) // 8
    if tag == _WAVE_FORMAT_IEEE_FLOAT:
        if width not in (4, 8):
            raise ValueError('unsupported float sample width %d' % width)
        return np.frombuffer(raw, dtype='<f4' if width == 4 else '<f8').astype(dtype)
    if width == 2:
        return np.frombuffer(raw, dtype='<i2').astype(dtype) / dtype(32768.0)
    if width == 4:
        return (np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0).astype(dtype)
    if width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        return (v - ((v & 0x800000) << 1)).astype(dtype) / dtype(8388608.0)
    if width == 1:
        return (np.frombuffer(raw, dtype=np.uint8).astype(dtype) - dtype(128.0)) / dtype(128.0)
    raise ValueError('unsupported sample width %d' % width)


def read_wav(path, start=0, stop=None, dtype=np.float64):
    """WAV -> (dtype [frames, channels] in [-1, 1), sample rate), with a partial read like
    ``soundfile.read(path, start=, stop=)`` (data/dataset.py:194): only the requested frames are read from disk.
    8/16/24/32-bit integer PCM and 32/64-bit float, plain or WAVE_FORMAT_EXTENSIBLE -- what MedleyDB / MUSDB18-HQ ship.
    dtype=np.float32 is lossless for integer PCM up to 24 bits (and halves the bytes sent to the GPU)."""
    h = wav_header(path)
    n = h['frames']
    stop = n if stop is None else min(stop, n)
    start = min(max(start, 0), stop)
    with open(path, 'rb') as fh:
        fh.seek(h['data_offset'] + start * h['frame_bytes'])
        raw = fh.read((stop - start) * h['frame_bytes'])
    return _decode(raw, h, np.dtype(dtype).type).reshape(-1, h['channels']), h['rate']


def read_wav_native(path, start=0, stop=None, out=None):
    """Partial read WITHOUT conversion, for the device front-end (features.stft_logmag on int16 / int32 tensors scales by
    1/2^(bits-1) itself, exactly as soundfile does): 16-bit PCM -> int16 [frames, channels]; 32-bit PCM -> int32; 24-bit PCM
    -> int32 left-justified (value << 8, so the same 1/2^31 scale applies); float32 files -> float32; anything else (8-bit,
    float64) -> float32 through read_wav.  out: optional preallocated array of the right dtype and shape (e.g. a view of a
    page-locked staging buffer) -- 16 / 32-bit samples are then read from the file straight into it.
    Returns (array, sample rate)."""
    h = wav_header(path)
    n = h['frames']
    stop = n if stop is None else min(stop, n)
    start = min(max(start, 0), stop)
    width, ch = (h['bits'] + 7) // 8, h['channels']
    kind = native_dtype(h)
    if kind is None:
        a, rate = read_wav(path, start, stop, dtype=np.float32)
        if out is not None:
            out[...] = a
            return out, rate
        return a, rate
    if width != 3 and out is not None and out.dtype == kind and out.shape == (stop - start, ch) and out.flags.c_contiguous:
        # the ingest path: positioned read on a cached descriptor straight into the caller's (page-locked) buffer -- no
        # open / seek / close per chunk, no file position shared between the decode threads
        fd, cached = _get_fd(path, h)
        try:
            buf, off, want = memoryview(out).cast('B'), h['data_offset'] + start * h['frame_bytes'], out.nbytes
            got = 0
            while got < want:
                r = os.preadv(fd, [buf[got:]], off + got)
                if r <= 0:
                    raise ValueError('%s: short read' % path)
                got += r
        finally:
            _put_fd(fd, cached)
        return out, h['rate']
    with open(path, 'rb') as fh:
        fh.seek(h['data_offset'] + start * h['frame_bytes'])
        if width != 3:
            a = np.frombuffer(fh.read((stop - start) * h['frame_bytes']), dtype=kind).reshape(-1, ch)
        else:
            b = np.frombuffer(fh.read((stop - start) * h['frame_bytes']), dtype=np.uint8).reshape(-1, 3)
            a = np.zeros((b.shape[0], 4), dtype=np.uint8)
            a[:, 1:] = b                                       # little endian: byte 0 of the int32 stays zero = value << 8
            a = a.view('<i4').reshape(-1, ch)
    if out is not None:
        out[...] = a
        return out, h['rate']
    return a, h['rate']


def native_dtype(h):
    """numpy dtype read_wav_native yields for a parsed header (None: converted to float32 on the host)."""
    width = (h['bits'] + 7) // 8
    if h['tag'] == _WAVE_FORMAT_IEEE_FLOAT:
        return np.dtype('<f4') if width == 4 else None
    return {2: np.dtype('<i2'), 3: np.dtype('<i4'), 4: np.dtype('<i4')}.get(width)


def wav_num_frames(path):
    h = wav_header(path)
    return h['frames'], h['rate']


_resample_warned = False


def resample(a, rate, sr):
    """[frames, channels] at `rate` Hz -> `sr` Hz, float32: what ``librosa.load(path, sr=sr)`` does on a rate mismatch
    (data/dataset_utils.py:53-83).  librosa's resampler is a third-party dependency that is neither vendored nor installed
    (soxr_hq in librosa >= 0.10, resampy's kaiser_best before): parity with it is UNPINNED -- this is scipy's polyphase
    Kaiser-windowed FIR (scipy.signal.resample_poly), the same class of band-limited interpolator, output length
    ceil(n * sr / rate) as librosa's.  MedleyDB / MUSDB18-HQ are 44.1 kHz throughout, so no reference caller resamples."""
    global _resample_warned
    from fractions import Fraction
    from scipy.signal import resample_poly
    if not _resample_warned:
        import warnings
        warnings.warn('resampling %d Hz audio to %d Hz on load with scipy.signal.resample_poly (librosa.load would use soxr / '
                      'resampy: samples agree to the resamplers\' pass-band ripple, not bit for bit)' % (rate, sr), RuntimeWarning,
                      stacklevel=3)
        _resample_warned = True
    f = Fraction(int(sr), int(rate))
    out = resample_poly(a.astype(np.float64), f.numerator, f.denominator, axis=0)
    n = -(-a.shape[0] * int(sr) // int(rate))
    return np.ascontiguousarray(out[:n]).astype(np.float32)


def _load(path, sr):
    a, rate = read_wav(path, dtype=np.float32)             # librosa.load yields float32
    if sr is not None and rate != sr:
        a = resample(a, rate, sr)
    return a[:, 0].copy() if a.shape[1] == 1 else a.T.copy()   # mono=False: [n] for mono files, else [channels, n]


def load_tracks(base_dir, song_name, tracklist=('bass', 'drums', 'vocals', 'other', 'mix'), sr=44100) -> dict:
    """data/dataset_utils.py:53-68, MedleyDB layout: {track: float32 ndarray[channels, n]} (what
    ``librosa.load(path, sr=sr, mono=False)`` returns for files already at `sr`)."""
    out = {}
    for track in tracklist:
        if track == 'mix':
            p = os.path.join(base_dir, song_name, '{}_MIX.wav'.format(song_name))
        else:
            p = os.path.join(base_dir, song_name, '{}_STEMS_JOINED'.format(song_name),
                             '{}_STEM_{}.wav'.format(song_name, track.upper()))
        out[track] = _load(p, sr)
    return out


def load_tracks_musdb18(base_dir, song_name, tracklist=('bass', 'drums', 'vocals', 'other', 'mix'), sr=44100) -> dict:
    """data/dataset_utils.py:71-83, MUSDB18-HQ layout: {song}/{bass,drums,vocals,other,mixture}.wav."""
    return {track: _load(os.path.join(base_dir, song_name, '{}.wav'.format('mixture' if track == 'mix' else track)), sr)
            for track in tracklist}

"""MultitrackAudioDataset -- drop-in for the reference's data/dataset.py:16-304 with the feature front-end on
the GPU (dam_stft_logmag_f32).

Same constructor, ``__getitem__(i) -> (train_features [S,1025,T], gt_features [1025,T])``, ``__len__``,
``compute_features(audio, window_size=2048, hop_length=1024)``, ``get_tracklist`` / ``get_num_songs`` /
``get_song_durations`` and the same chunk <-> song index arithmetic (:97-113).  Differences, all explicit:
  * items are float32 CUDA tensors (the reference yields float64 CPU tensors that its own ModelTrainer cannot
    consume) when ``__getitem__`` runs in the process that owns the GPU;
  * inside a DataLoader WORKER (``DataLoader(d_train, batch_size=48, num_workers=6, pin_memory=True)``, training.ipynb
    cell 6 as written) ``__getitem__`` touches no GPU API: it returns the decoded chunk as a ``PcmItem`` (host PCM, the
    file's own sample type); the default collate turns a list of them into ONE ``HostPcmBatch`` ([B, S+1, n, ch], in
    shared memory), the loader's pin-memory thread page-locks it, and in the main process the batch either goes to
    ``ModelTrainer`` (upload on a copy stream + the PCM-fed captured step: the front-end runs inside the graph) or is
    unpacked like the reference's ``(train_features, gt_features)`` pair -- the upload and ONE front-end launch then happen
    on the spot (``features.batch_features(batch)`` is the same as a function);
  * audio comes from MedleyDB-layout WAV files read with the stdlib (``soundfile`` is not required), or from
    in-memory arrays via ``MultitrackAudioDataset.from_arrays``;
  * the per-item prints of the reference (:284,287-289) are behind ``verbose=True``;
  * ``normalize=True`` enables the per-frame max-abs normalisation that is commented out at :159-160;
  * ``compute_features=False`` reads the pre-computed feature cache written by ``_precompute_features()``; reader and
    writer agree on the writer's file names (they do not at the reference HEAD);
  * ``tracklist`` may name any number of stems (last entry = the target mix);
  * ``iter_batches(batch_size)`` is the batched ingest path: WAV chunks are decoded by a thread pool straight into
    page-locked staging buffers, travel on a copy stream while the previous batch is in use, and all B*(S+1) tracks of
    a batch go through ONE front-end launch (the per-item ``__getitem__`` + DataLoader path stays for drop-in use).
"""
import os
import random
import time
import weakref

import numpy as np
import torch
from torch.utils import data

from .. import features, staging
from .dataset_utils import (cached_files_begin, close_cached_files, native_dtype, read_wav, read_wav_native, wav_header,
                            wav_num_frames)


def in_loader_worker():
    """True where no GPU API may be touched: a DataLoader worker process, or any fork of a GPU-initialised parent (the HIP
    runtime does not survive a fork: torch refuses to re-initialise it, and a raw HIP call there would hang or fault)."""
    return data.get_worker_info() is not None or torch.cuda._is_in_bad_fork()


# datasets alive in THIS process by their token: a HostPcmBatch that comes back from a worker finds the dataset object
# the main process holds (the workers' copies are forks whose read counters die with them)
_live_datasets = weakref.WeakValueDictionary()
# a DataLoader over one of them forks its workers from the GPU-owning process at the start of every epoch: page-locked host
# buffers are taken out of what the children inherit (staging.install_fork_guard, include/dam_hip.h dam_host_dontfork_pinned)
staging.install_fork_guard(lambda: len(_live_datasets) > 0)


class MultitrackAudioDataset(data.Dataset):
    def __init__(self, base_path: str, songlist: list = None, chunk_length: int = 5, sr: int = 44100,
                 seed: int = None, normalize: bool = False, compute_features: bool = True,
                 augment_data: bool = False, *, tracklist=None, device=None, verbose=False, _arrays=None):
        self._base_path = base_path
        self._chunk_length = chunk_length
        self._normalize = normalize
        # staging.StepMark of the training step that consumes iter_batches (ModelTrainer arms it once its step is captured): the
        # feeder enqueues an upload only when the step enqueued last has reached its backward pass -- a host-to-device copy
        # beside the forward pass slows the forward's latency-bound launches (profiles/r05_pcie_trace.txt)
        self.upload_gate = None
        self._compute_features = compute_features
        self._augment = augment_data
        self._sr = sr
        self._tracklist = list(tracklist) if tracklist else ['bass', 'drums', 'vocals', 'other', 'mix']
        self._device = torch.device(device) if device is not None else torch.device('cuda')
        self._verbose = verbose
        self._arrays = _arrays
        # device-side augmentation draws are keyed by (seed, item, how often the item was read before, track): every access
        # is a fresh draw, as the reference's np.random.uniform per access (data/dataset.py:164-168), and a run is
        # reproducible given the seed; without a seed the base is random, like numpy's unseeded global state
        self._aug_seed = seed if seed else int.from_bytes(os.urandom(8), 'little') >> 2
        self._aug_reads, self._aug_epoch = {}, 0
        self._token = int.from_bytes(os.urandom(8), 'little') >> 1
        _live_datasets[self._token] = self
        if _arrays is not None and not songlist:
            songlist = list(_arrays.keys())
        if not songlist:
            songlist = [song_name for song_name in os.listdir(self._base_path)
                        if os.path.isdir(os.path.join(self._base_path, song_name))]
        self.songlist = songlist
        if seed:
            random.seed(seed)
        random.shuffle(songlist)
        self._len, self.song_durations = self._calculate_dataset_length()

    _AUG_READ_SHIFT = 40       # read counter folded into the item field of the draw's key (items < 2^40, reads < 4096)

    def set_epoch(self, epoch: int):
        """The next read of every item takes augmentation draw number `epoch` (resuming a run: set_epoch(k) before pass k
        reproduces that pass's gains).  Without it the counter simply advances with every read of an item."""
        self._aug_reads, self._aug_epoch = {}, int(epoch)

    def _aug_keys(self, items):
        """Global item indices -> the item keys of THIS access' draws (features.augment_gains(items=)); advances the
        items' read counters."""
        keys = []
        for i in items:
            k = self._aug_reads.get(i, self._aug_epoch)
            self._aug_reads[i] = k + 1
            keys.append(int(i) + ((k % 4096) << self._AUG_READ_SHIFT))
        return keys

    @classmethod
    def from_arrays(cls, songs: dict, **kwargs):
        """songs: {song_name: {track_name: ndarray[n, channels] or [n]}} held in memory (no disk)."""
        return cls(None, _arrays=songs, **kwargs)

    # ---- index arithmetic (data/dataset.py:56-75, 97-113)
    def _track_frames(self, song_name):
        if self._arrays is not None:
            return self._arrays[song_name][self._tracklist[-1]].shape[0], self._sr
        return wav_num_frames(self._get_track_path(song_name, self._tracklist[-1]))

    def _calculate_dataset_length(self) -> tuple:
        """data/dataset.py:56-75: whole seconds of the mix, trimmed to a multiple of the chunk length."""
        total, durations = 0, []
        for song_name in self.songlist:
            n, sr = self._track_frames(song_name)
            seconds = int(n / sr)
            durations.append(seconds - seconds % self._chunk_length)
            total += int(seconds / self._chunk_length)
        return total, durations

    def _get_track_path(self, song_name: str, track_name: str) -> str:
        if track_name == 'mix':
            return os.path.join(self._base_path, song_name, '{}_MIX.wav'.format(song_name))
        return os.path.join(self._base_path, song_name, '{}_STEMS_JOINED'.format(song_name),
                            '{}_STEM_{}.wav'.format(song_name, track_name.upper()))

    def _calculate_song_index(self, chunk_i: int) -> tuple:
        """Global chunk index -> (song, chunk inside the song): data/dataset.py:97-113 walks the songs one by one (per item, in
        every worker); here a bisection over the cumulative chunk counts.  Same answers, including the reference's behaviour
        past the end: indices beyond the last chunk land in the last song with a chunk number that runs on."""
        starts = getattr(self, '_chunk_starts', None)
        if starts is None or len(starts) != len(self.songlist):
            starts, total = [], 0
            for d in self.song_durations:
                starts.append(total)
                total += int(d / self._chunk_length)
            self._chunk_starts = starts
        from bisect import bisect_right
        song_i = max(0, min(bisect_right(starts, chunk_i) - 1, len(starts) - 1))
        return song_i, chunk_i - starts[song_i]

    # ---- features (data/dataset.py:132-162) on the GPU
    def compute_features(self, audio, window_size: int = 2048, hop_length: int = 1024):
        """audio: mono ndarray/tensor [n] (or [n, channels]: channels are averaged first).
        Returns the dB spectrogram, float32 CUDA tensor [window_size/2+1, 1 + n // hop_length]."""
        a = torch.as_tensor(audio)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float64)
        a = a.to(self._device)
        return features.stft_logmag(a[None], window_size, hop_length, normalize=self._normalize)[0]

    @staticmethod
    def _augment_audio(audio, gain_from: float = 0.6, gain_to: float = 1.4):
        """data/dataset.py:164-168."""
        return np.random.uniform(gain_from, gain_to) * audio

    @staticmethod
    def _augment_features(feats, gain_from: float = 0.6, gain_to: float = 1.4):
        """data/dataset.py:170-179: one dB offset per stem."""
        gains_db = 20 * np.log10(np.random.uniform(gain_from, gain_to, size=len(feats)))
        return feats + torch.as_tensor(gains_db, dtype=feats.dtype, device=feats.device)[:, None, None]

    @staticmethod
    def _stereo_to_mono(audio: np.ndarray) -> np.ndarray:
        """data/dataset.py:181-183 (the GPU path fuses this mean into the STFT kernel's load)."""
        return np.mean(audio, axis=1)

    def _read_chunk(self, song_name, track_name, lo, hi):
        """One track's chunk as the front-end wants it: the file's own integer samples where the kernel reads them (16 / 24 /
        32-bit PCM -> int16 / int32, dataset_utils.read_wav_native: no host conversion), in-memory arrays as they are."""
        if self._arrays is not None:
            return np.asarray(self._arrays[song_name][track_name][lo:hi])
        path = self._get_track_path(song_name, track_name)
        if native_dtype(wav_header(path)) is None:
            # 8-bit PCM / float64 files: no sample type the kernel reads -- float64 in [-1, 1), what soundfile.read yields
            # at data/dataset.py:194 (the front-end takes float64 PCM as it is)
            return read_wav(path, lo, hi, dtype=np.float64)[0]
        return read_wav_native(path, lo, hi)[0]

    @staticmethod
    def _common_pcm(chunks):
        """Tracks of one item stacked into ONE front-end input: their common dtype if they share one, else float64 in
        [-1, 1) (what soundfile.read yields at data/dataset.py:194)."""
        kinds = {c.dtype for c in chunks}
        if len(kinds) == 1 and chunks[0].dtype in (np.int16, np.int32, np.float32, np.float64):
            return np.stack(chunks)
        out = []
        for c in chunks:
            if c.dtype == np.int16:
                c = c.astype(np.float64) / 32768.0
            elif c.dtype == np.int32:
                c = c.astype(np.float64) / 2147483648.0
            out.append(c.astype(np.float64))
        return np.stack(out)

    def _read_item_pcm(self, song_i: int, chunk_i: int):
        """data/dataset.py:192-196: the S+1 tracks of one chunk as ONE host array [S+1, n, channels] (their common sample
        type).  Host work only -- this is all a DataLoader worker does for an item."""
        song_name = self.songlist[song_i]
        lo, hi = chunk_i * self._chunk_length * self._sr, (chunk_i + 1) * self._chunk_length * self._sr
        chunks = [self._read_chunk(song_name, t, lo, hi) for t in self._tracklist]
        chunks = [c[:, None] if c.ndim == 1 else c for c in chunks]
        return self._common_pcm(chunks)

    def _global_index(self, song_i, chunk_i):
        return sum(int(d / self._chunk_length) for d in self.song_durations[:song_i]) + chunk_i

    def _process_on_the_fly(self, song_i: int, chunk_i: int, index: int = None) -> tuple:
        """data/dataset.py:185-210: all S+1 tracks of the chunk go through ONE front-end launch."""
        pcm = torch.from_numpy(self._read_item_pcm(song_i, chunk_i)).to(self._device)          # [S+1, n, channels]
        gain = None
        if self._augment:        # one draw per track, the mix included (data/dataset.py:198-199): drawn on the device,
            # reproducibly, keyed by (dataset seed, global item index, read count, track) instead of numpy's global state
            item = index if index is not None else self._global_index(song_i, chunk_i)
            gain = features.augment_gains(self._aug_seed, pcm.shape[0], items=self._aug_keys([item]), device=self._device)[0]
        feats = features.stft_logmag(pcm, 2048, 1024, gain=gain, normalize=self._normalize)
        return DeviceItem((feats[:-1], feats[-1]))

    # ---- batched ingest: decode threads -> page-locked staging -> copy stream -> one front-end launch per batch
    def staging_format(self):
        """(numpy dtype, channels) of the page-locked staging buffers of iter_batches: the files' own sample type when every
        track of every song shares it (16-bit MedleyDB / MUSDB18-HQ stems travel as int16: half the PCIe bytes of float32
        and no conversion on the host), else float32."""
        if self._arrays is not None:      # in-memory songs: their own sample type when all share one the kernel reads
            arrs = [np.asarray(self._arrays[song][t]) for song in self.songlist for t in self._tracklist]
            kinds = {a.dtype for a in arrs}
            kind = kinds.pop() if len(kinds) == 1 else None
            a = arrs[0]
            ok = kind in (np.dtype(np.int16), np.dtype(np.int32), np.dtype(np.float32))
            return (kind if ok else np.dtype(np.float32)), (1 if a.ndim == 1 else a.shape[1])
        kinds, chans = set(), set()
        for song in self.songlist:
            for t in self._tracklist:
                h = wav_header(self._get_track_path(song, t))
                kinds.add(native_dtype(h))
                chans.add(h['channels'])
        if len(chans) != 1:
            raise ValueError('iter_batches needs one channel count over all tracks (found %s)' % sorted(chans))
        kind = kinds.pop() if len(kinds) == 1 else None
        return (np.dtype(np.float32) if kind is None else kind), chans.pop()

    def iter_batches(self, batch_size, indices=None, workers=8, drop_last=False, pcm=False):
        """pcm=True: yields PcmBatch objects instead -- the uploaded PCM of the batch ([B, S+1, n, ch] in the staging format)
        and its augmentation gains, NOT yet turned into features: ModelTrainer binds them to a step whose captured graph
        contains the front-end (engine.TrainStep.bind_clips: no feature copy, no separate front-end launch); any other
        consumer unpacks them like a (features, target) pair and gets the features computed on the spot.

        Yields (train_features [B,S,1025,T], gt_features [B,1025,T]) float32 CUDA tensors for consecutive groups of
        `batch_size` items of `indices` (default: every item, in order) -- what ``DataLoader(self, batch_size)`` yields,
        with the reads of data/dataset.py:192-196 done by `workers` threads straight into page-locked memory (integer PCM
        stays integer: staging_format), three staging slots -- batch k in use, k+1 on the copy stream, k+2 being decoded in
        the background (also while the consumer waits in ``loss.item()``) -- and the augmentation gains
        (data/dataset.py:198-199) drawn on the device per (seed, item, read count, track)."""
        from concurrent.futures import ThreadPoolExecutor
        idx = list(range(len(self))) if indices is None else [int(i) for i in indices]
        groups = [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
        if drop_last and groups and len(groups[-1]) < batch_size:
            groups.pop()
        if not groups:
            return
        K, n = len(self._tracklist), self._chunk_length * self._sr
        kind, ch = self.staging_format()
        tdt = torch.from_numpy(np.empty(0, dtype=kind)).dtype
        NS = 3            # staging slots: batch j in use, j + 1 travelling, j + 2 being decoded
        # page-locking 3 x 38-76 MB costs milliseconds: the staging buffers are kept on the dataset and reused by the next pass
        # (one epoch = one iter_batches call; a pass that is still running keeps its own set)
        key = (batch_size, K, n, ch, tdt, str(self._device))
        cache = getattr(self, '_staging_cache', None)
        if cache is not None and cache[0] == key and not cache[3].locked():
            host, dev, busy = cache[1], cache[2], cache[3]
        else:
            import threading as _th
            host = [torch.empty((batch_size, K, n, ch), dtype=tdt, pin_memory=True) for _ in range(NS)]
            dev = [torch.empty((batch_size, K, n, ch), dtype=tdt, device=self._device) for _ in range(NS)]
            busy = _th.Lock()
            self._staging_cache = (key, host, dev, busy)
        import threading
        busy.acquire()
        cached_files_begin()
        uploaded = [torch.cuda.Event() for _ in range(NS)]
        # pcm=True: the consumer's step reads dev[slot] in place.  The HOST half of a slot is free as soon as its upload has left it
        # (the feeder reads the next batch into it at once, as in feature mode); only the upload INTO dev[slot] waits -- in the
        # feeder thread -- for what the consumer enqueued on it: consumed[slot] is recorded when the consumer comes back
        # for its next batch, and consumed_set[slot] tells the feeder that this recording has happened
        consumed = [torch.cuda.Event() for _ in range(NS)]
        consumed_set = [threading.Event() for _ in range(NS)]
        for ev in consumed_set:
            ev.set()
        copy_stream = torch.cuda.Stream(device=self._device)

        import queue
        ready, free, stop = queue.Queue(), threading.Semaphore(NS), threading.Event()

        def feeder(pool):
            """Producer thread: for batch j = 0, 1, ... take a free staging slot, read the batch into host[slot] (pool
            threads), enqueue its upload on the copy stream the moment the last read is done, hand (slot, group) over.
            None of this is on the consumer's critical path: it runs while the consumer's step executes / syncs."""
            from collections import deque
            started = deque()                 # batches whose reads are running: (slot, group, futures)
            try:
                j = 0
                while j < len(groups) or started:
                    # start the reads of as many batches as there are free slots (block for a slot only when idle)
                    while j < len(groups) and free.acquire(blocking=not started):
                        if stop.is_set():
                            return
                        slot, group = j % NS, groups[j]
                        with staging.capture_guard:
                            uploaded[slot].synchronize()                # the slot's previous upload has left host[slot]
                        view = host[slot].numpy()

                        # one task = a few tracks of one clip (K tracks in `per` pieces): a task per track made the pool's
                        # bookkeeping (72 futures per batch, all under the interpreter lock) a third of a batch's time
                        per = max(1, min(K, (len(group) * K + 2 * max(1, workers) - 1) // (2 * max(1, workers))))

                        def one(job, view=view):
                            b, k0, k1, item = job
                            song_i, chunk_i = self._calculate_song_index(item)
                            for k in range(k0, k1):
                                self._read_chunk_into(view[b, k], self.songlist[song_i], self._tracklist[k], chunk_i * n, (chunk_i + 1) * n)
                        started.append((slot, group, [pool.submit(one, (b, k0, min(K, k0 + per), item))
                                                      for b, item in enumerate(group) for k0 in range(0, K, per)]))
                        j += 1
                    slot, group, reads = started.popleft()
                    for f in reads:
                        f.result()                                      # raises a reader's error (re-raised in the consumer)
                    if stop.is_set():
                        return
                    B = len(group)
                    # upload AND front-end launch on the copy stream, from this thread: the features of batch j + 1 are
                    # computed while the consumer's step on batch j runs -- nothing of it is left on the consumer's
                    # critical path (with a per-step loss.item() the device idles for every host call in between).
                    # capture_guard: never while the consumer captures its step into a hipGraph (ModelTrainer, third batch)
                    if pcm:
                        while not consumed_set[slot].wait(0.05):                    # (normally set long ago: three batches back)
                            if stop.is_set():
                                return
                        consumed_set[slot].clear()
                        # the consumer's step read dev[slot] in place: it must be over before the upload overwrites it.  THIS
                        # thread waits, not the copy stream: a stream that waits for an event of the training stream costs the
                        # training stream 0.09 ms per step on this stack (profiles/r05_sync_cost_probe.txt), a host wait nothing
                        consumed[slot].synchronize()
                    gate = self.upload_gate
                    if gate is not None:
                        gate.synchronize()          # (a step enqueued BEFORE this batch was handed over: it cannot wait for us)
                    with staging.capture_guard, torch.cuda.device(self._device), torch.cuda.stream(copy_stream):
                        dev[slot][:B].copy_(host[slot][:B], non_blocking=True)
                        uploaded[slot].record(copy_stream)
                        gain = None
                        if self._augment:
                            gain = features.augment_gains(self._aug_seed, K, items=self._aug_keys(group), device=self._device)
                        if pcm:
                            payload = PcmBatch(dev[slot][:B], gain, self._normalize)
                        else:
                            payload = features.stft_logmag_clips(dev[slot][:B], 2048, 1024, gain=gain, normalize=self._normalize)
                        done = torch.cuda.Event()
                        done.record(copy_stream)
                    ready.put((payload, done, slot))
                ready.put(None)
            except BaseException as e:       # noqa: B036 -- handed to the consumer, which re-raises it
                ready.put(e)
            finally:
                # an early exit (the consumer stopped iterating, or a reader raised) leaves reads running that write into the
                # page-locked slots: they are waited for HERE, before the consumer's `finally` may hand the slots to the next
                # pass (a cancelled future never ran; a running one is joined)
                for _, _, reads in started:
                    for f in reads:
                        if not f.cancel():
                            try:
                                f.result()
                            except BaseException:      # noqa: B036 -- already reported, or irrelevant after a stop
                                pass

        try:
            with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
                th = threading.Thread(target=feeder, args=(pool,), daemon=True)
                th.start()
                try:
                    while True:
                        item = ready.get()
                        if item is None:
                            break
                        if isinstance(item, BaseException):
                            raise item
                        payload, done, slot = item
                        cur = torch.cuda.current_stream(self._device)
                        cur.wait_event(done)
                        if pcm:
                            if payload.gain is not None:
                                payload.gain.record_stream(cur)
                            free.release()                  # the host half of the slot: its upload has been enqueued
                            try:
                                yield payload
                            finally:
                                # back here the consumer has enqueued everything that reads this batch (ModelTrainer: the
                                # step's graph replay): the slot's next upload waits for that work on the device
                                consumed[slot].record(torch.cuda.current_stream(self._device))
                                consumed_set[slot].set()
                            continue
                        x, gt = payload
                        x.record_stream(cur)            # allocated on the copy stream, used on the consumer's
                        gt.record_stream(cur)
                        free.release()                  # the staging slot is free: its PCM has been turned into features
                        yield x, gt
                finally:
                    stop.set()
                    free.release()          # a feeder waiting for a slot sees `stop` and leaves
                    th.join()               # (it has joined its outstanding reads by then)
                    torch.cuda.current_stream(self._device).wait_stream(copy_stream)
        finally:
            # only now -- the pool has shut down, no reader can still write into host[] -- may another pass reuse the slots
            busy.release()
            close_cached_files()

    def batch_loader(self, batch_size, indices=None, workers=8, drop_last=False, pcm=False):
        """iter_batches as a re-iterable loader with ``len()`` -- what ModelTrainer.fit / the notebooks' loops expect of a
        DataLoader (one pass per ``for`` loop).  pcm=True: PcmBatch items (see iter_batches) -- ModelTrainer then runs the
        front-end inside its captured step."""
        return _BatchLoader(self, batch_size, indices, workers, drop_last, pcm)

    def _read_chunk_into(self, out, song_name, track_name, lo, hi):
        """One track's chunk into a [n, channels] view of a staging buffer (dtype: staging_format)."""
        if self._arrays is not None:
            a = np.asarray(self._arrays[song_name][track_name][lo:hi])
            out[...] = a[:, None] if a.ndim == 1 else a
        elif out.dtype == np.float32:
            out[...] = read_wav(self._get_track_path(song_name, track_name), lo, hi, dtype=np.float32)[0]
        else:
            read_wav_native(self._get_track_path(song_name, track_name), lo, hi, out=out)

    # ---- pre-computed feature cache (data/dataset.py:213-268).  The reference's writer emits
    #      {song}_FEATURES/{i}_train_{len}s[_norm].npy / {i}_gt_{len}s[_norm].npy while its reader looks for
    #      {i}_train[_norm].npy: here both sides use the WRITER's names.
    def _feature_paths(self, song_i, chunk_i):
        song = self.songlist[song_i]
        d = os.path.join(self._base_path, song, '{}_FEATURES'.format(song))
        suffix = '_norm' if self._normalize else ''
        return (d, os.path.join(d, '{}_train_{}s{}.npy'.format(chunk_i, self._chunk_length, suffix)),
                os.path.join(d, '{}_gt_{}s{}.npy'.format(chunk_i, self._chunk_length, suffix)))

    def _precompute_features(self):
        """Runs the GPU front-end over every chunk of every song and stores float32 .npy files next to the audio."""
        if self._arrays is not None:
            raise ValueError('the feature cache lives next to the audio files: needs a base_path dataset')
        for song_i in range(len(self.songlist)):
            for chunk_i in range(int(self.song_durations[song_i] / self._chunk_length)):
                d, p_train, p_gt = self._feature_paths(song_i, chunk_i)
                os.makedirs(d, exist_ok=True)
                aug, self._augment = self._augment, False          # the cache holds un-augmented features
                try:
                    train, gt = self._process_on_the_fly(song_i, chunk_i)
                finally:
                    self._augment = aug
                np.save(p_train, train.cpu().numpy())
                np.save(p_gt, gt.cpu().numpy())

    def _process_precomputed(self, song_i, chunk_i) -> tuple:
        _, p_train, p_gt = self._feature_paths(song_i, chunk_i)
        train, gt = torch.from_numpy(np.load(p_train)), torch.from_numpy(np.load(p_gt))
        if in_loader_worker():       # a DataLoader worker: CPU tensors (default collate stacks them, the pin thread page-locks
            # them, ModelTrainer's / the caller's .to(device) uploads them -- the reference's own arrangement)
            return (MultitrackAudioDataset._augment_features(train) if self._augment else train), gt
        train, gt = train.to(self._device), gt.to(self._device)
        if self._augment:
            train = MultitrackAudioDataset._augment_features(train)
        return DeviceItem((train, gt))

    def __getitem__(self, index: int) -> tuple:
        song_i, chunk_i = self._calculate_song_index(index)
        if self._verbose:
            print('Song {}, chunk {}'.format(self.songlist[song_i], chunk_i))
        if not self._compute_features:
            return self._process_precomputed(song_i, chunk_i)
        if in_loader_worker():
            # training.ipynb cell 6: DataLoader(d_train, num_workers=6, pin_memory=True).  No GPU API here: the decoded
            # chunk travels as host PCM; the augmentation draw (keyed by item and read count) and the front-end happen
            # in the process that owns the GPU (HostPcmBatch)
            return PcmItem(torch.from_numpy(self._read_item_pcm(song_i, chunk_i)), int(index), self._token,
                           self._aug_seed if self._augment else None, self._normalize, str(self._device))
        tic = time.time()
        item = self._process_on_the_fly(song_i, chunk_i, index)          # (train_features, gt_features)
        if self._verbose:
            print('Features: {}'.format(time.time() - tic))
        return item

    def __getitems__(self, indices):
        """What a DataLoader's fetcher calls with the indices of one batch (torch >= 2.0).  In a worker the whole batch is
        decoded into ONE shared-memory block [B, S+1, n, ch] -- every track read straight to its place, as iter_batches
        reads into its page-locked slots: no per-item stack, no batch stack (a C3 batch is 38 MB; the two copies were two
        thirds of a worker's time) -- and the items are its rows; elsewhere, and for anything the block form does not cover
        (mixed channel counts, the feature cache), the items one by one."""
        if not (self._compute_features and in_loader_worker()):
            return [self[i] for i in indices]
        try:
            fmt = getattr(self, '_worker_format', None)
            if fmt is None:
                fmt = self._worker_format = self.staging_format()
        except ValueError:
            return [self[i] for i in indices]
        kind, ch = fmt
        K, n = len(self._tracklist), self._chunk_length * self._sr
        elem = torch.from_numpy(np.empty(0, dtype=kind))
        shape = (len(indices), K, n, ch)
        block, state = self._worker_block(elem, shape)
        view = block.numpy()
        items = []
        for b, index in enumerate(indices):
            song_i, chunk_i = self._calculate_song_index(index)
            if self._verbose:
                print('Song {}, chunk {}'.format(self.songlist[song_i], chunk_i))
            for k, track in enumerate(self._tracklist):
                self._read_chunk_into(view[b, k], self.songlist[song_i], track, chunk_i * n, (chunk_i + 1) * n)
            items.append(PcmItem(block[b], int(index), self._token, self._aug_seed if self._augment else None, self._normalize,
                                 str(self._device), block=(block, b, state)))
        return items

    SHM_RING = 3          # reusable shared-memory blocks per worker: two prefetched batches + the one the pin thread is copying

    def _worker_block(self, elem, shape):
        """-> (block [B, S+1, n, ch], state): where a worker decodes a batch.  A FRESH shared-memory segment per batch is what
        torch's own collate allocates -- and costs a worker more than the decoding: 38 MB = 9,300 first-touch page faults
        (allocate, zero, charge) per C3 batch, ~40 ms, i.e. six workers deliver a batch every 8 ms to a step that takes 4.3
        (profiles/r05_dataloader_workers.txt).  So each worker keeps SHM_RING blocks and re-uses one once the consumer has
        said it is done with it: `state` is a shared int32 word, 1 while the batch is in flight, set back to 0 by
        HostPcmBatch.pin_memory() after it has copied the block into page-locked memory (the loader's pin thread:
        pin_memory=True, the notebooks' setting).  A consumer that never pins never releases: the ring's blocks stay with
        their batches and every further batch gets a fresh segment, as before (state None)."""
        if data.get_worker_info() is None:
            return torch.empty(shape, dtype=elem.dtype), None

        def shared(e, sh):
            numel = 1
            for d in sh:
                numel *= d
            return e.new(e._typed_storage()._new_shared(numel, device=e.device)).resize_(*sh)
        ring = self.__dict__.setdefault('_shm_ring', [])
        if ring and (tuple(ring[0][0].shape) != tuple(shape) or ring[0][0].dtype != elem.dtype):
            return shared(elem, shape), None                      # another shape (a ragged last batch): a one-off segment
        for block, state in ring:
            if int(state[0]) == 0:
                state[0] = 1
                return block, state
        if len(ring) < self.SHM_RING:
            state = shared(torch.empty(0, dtype=torch.int32), (1,))
            state[0] = 1
            ring.append((shared(elem, shape), state))
            return ring[-1]
        return shared(elem, shape), None

    def __len__(self) -> int:
        return self._len

    def __getstate__(self):
        """Pickled for DataLoader workers started by spawn / forkserver: the page-locked staging slots and their lock stay
        behind (iter_batches rebuilds them)."""
        state = dict(self.__dict__)
        state.pop('_staging_cache', None)
        state.pop('_worker_format', None)
        state.pop('_shm_ring', None)
        return state

    def get_num_songs(self) -> int:
        return len(self.songlist)

    def get_song_durations(self) -> list:
        return self.song_durations

    def get_tracklist(self) -> list:
        return self._tracklist

    def compute_mean_loudness(self) -> dict:
        """data/dataset.py:115-130: mean BS.1770 integrated loudness of every stem over the song list (whole files).
        The meter is the HIP one (loudness.Meter); same progress lines as the reference."""
        from statistics import mean
        from ..loudness import Meter
        print('[.] Computing mean loudness...')
        loudness = {track_name: [] for track_name in self._tracklist}
        meter = Meter(self._sr)
        for song_i, song_name in enumerate(self.songlist):
            print('{}/{}: {}'.format(song_i + 1, len(self.songlist), song_name))
            for track_name in self._tracklist:
                if self._arrays is not None:
                    track = np.asarray(self._arrays[song_name][track_name])
                else:
                    track = read_wav(self._get_track_path(song_name, track_name))[0]
                loudness[track_name].append(meter.integrated_loudness(track))
        return {track_name: mean(loudness[track_name]) for track_name in loudness}


class PcmBatch:
    """One batch of uploaded clips that has not been through the front-end yet: `clips` [B, S+1, n, ch] on the device (mix
    last; float32, or int16 / int32 as the files hold it), `gain` [B, S+1] augmentation draws or None.  Unpacks like the
    (train_features, gt_features) pair of data/dataset.py:207-210 -- the front-end then runs on the current stream -- so a
    plain ``for feats, target in loader`` loop works; ModelTrainer takes the PCM itself (engine.TrainStep.bind_clips).
    Valid until the loader is asked for its next batch (the staging slot is then refilled)."""
    __slots__ = ('clips', 'gain', 'normalize', 'n_fft', 'hop')

    def __init__(self, clips, gain=None, normalize=False, n_fft=2048, hop=1024):
        self.clips, self.gain, self.normalize, self.n_fft, self.hop = clips, gain, bool(normalize), n_fft, hop

    def features(self):
        return features.stft_logmag_clips(self.clips, self.n_fft, self.hop, gain=self.gain, normalize=self.normalize)

    def __iter__(self):
        return iter(self.features())


class DeviceItem(tuple):
    """``(train_features, gt_features)`` as ``__getitem__`` yields them in the process that owns the GPU (data/dataset.py:292):
    a plain 2-tuple of CUDA tensors to every caller.  Its own type only so that the default collate produces a DeviceBatch,
    which ``DataLoader(..., num_workers=0, pin_memory=True)`` leaves alone (torch's pin step raises on CUDA tensors)."""
    __slots__ = ()


class DeviceBatch:
    """The collated ``(train_features [B,S,F,T], gt_features [B,F,T])`` pair of DeviceItems: unpacks, indexes and measures
    like the 2-tuple the default collate would have made; ``pin_memory()`` is the identity (the tensors are in HBM)."""
    __slots__ = ('pair',)

    def __init__(self, train_features, gt_features):
        self.pair = (train_features, gt_features)

    def __iter__(self):
        return iter(self.pair)

    def __getitem__(self, i):
        return self.pair[i]

    def __len__(self):
        return 2

    def pin_memory(self):
        return self


class PcmItem:
    """What ``__getitem__`` returns inside a DataLoader worker: one item's decoded chunk on the HOST -- `pcm` [S+1, n, ch]
    (mix last; the file's own sample type, data/dataset.py:192-196 without the float64 conversion), the item's global index
    (the key of its augmentation draw), the dataset's token / augmentation seed (None: no augmentation) / normalise flag /
    device.  A batch of them collates into a HostPcmBatch (registered with torch's default collate below)."""
    __slots__ = ('pcm', 'index', 'token', 'aug_seed', 'normalize', 'device', 'block')

    def __init__(self, pcm, index, token, aug_seed, normalize, device, block=None):
        self.pcm, self.index, self.token, self.aug_seed, self.normalize, self.device = pcm, index, token, aug_seed, normalize, device
        # (batch tensor, position): `pcm` is row `position` of a [B, S+1, n, ch] block that __getitems__ decoded the whole
        # batch into (shared memory) -- the collate then hands the block on as it is
        self.block = block

    def __iter__(self):
        """(train_features [S,F,T], gt_features [F,T]) -- in the process that owns the GPU; raises in a worker."""
        x, gt = collate_pcm_items([self]).features()
        return iter((x[0], gt[0]))


def _as_float64(pcm):
    if pcm.dtype == torch.int16:
        return pcm.to(torch.float64) / 32768.0
    if pcm.dtype == torch.int32:
        return pcm.to(torch.float64) / 2147483648.0
    return pcm.to(torch.float64)


def collate_pcm_items(batch, *, collate_fn_map=None):
    """torch.utils.data.default_collate for a list of PcmItems -> ONE HostPcmBatch.  Inside a worker the [B, S+1, n, ch]
    block is allocated in shared memory (what torch's own tensor collate does), so the batch crosses the process boundary
    as a file descriptor, not as a pickled copy."""
    first = batch[0]
    if first.block is not None and first.block[0].shape[0] == len(batch) and all(
            b.block is not None and b.block[0] is first.block[0] and b.block[1] == i for i, b in enumerate(batch)):
        # __getitems__ decoded this very batch straight into one (shared-memory) block: nothing to stack
        return HostPcmBatch(first.block[0], torch.tensor([b.index for b in batch], dtype=torch.int64), first.token,
                            first.aug_seed, first.normalize, first.device, release=first.block[2])
    pcms = [b.pcm for b in batch]
    if len({p.dtype for p in pcms}) > 1:         # files of different sample types in one batch: float64 in [-1, 1),
        pcms = [_as_float64(p) for p in pcms]    # what soundfile.read yields at data/dataset.py:194
    elem = pcms[0]
    if any(p.shape != elem.shape for p in pcms):
        raise ValueError('items of one batch must have one shape (chunk length x channels): %s'
                         % sorted({tuple(p.shape) for p in pcms}))
    out = None
    if data.get_worker_info() is not None:
        storage = elem._typed_storage()._new_shared(elem.numel() * len(pcms), device=elem.device)
        out = elem.new(storage).resize_(len(pcms), *elem.shape)
    clips = torch.stack(pcms, 0, out=out)
    return HostPcmBatch(clips, torch.tensor([b.index for b in batch], dtype=torch.int64), first.token, first.aug_seed,
                        first.normalize, first.device)


# The workers' re-usable shared blocks (MultitrackAudioDataset._worker_block) arrive in the main process as tensors rebuilt from a
# file descriptor: torch maps the block anew unless a storage of the same block is still ALIVE here (reductions.shared_cache holds
# weak references).  The batch object is dropped as soon as it has been copied into page-locked memory, so every batch was a fresh
# 38 MB mapping whose 9,300 pages the pin thread's copy then faulted in one by one -- 5 ms per batch in the loader's single pin
# thread, i.e. DataLoader(num_workers=6, pin_memory=True) delivered a batch every 4.8-6.3 ms (1.2 ms with pin_memory=False) to a
# step that takes 4.2 (tools/dl_worker_probe.py, profiles/r05_dataloader_pin_thread.txt).  Holding the last few dozen block tensors
# keeps their mappings (and page tables) alive: the next batch in the same block is found in torch's cache.  Bounded: blocks of
# finished epochs' workers fall out as new ones arrive (18 live blocks with six workers; a block is 38 MB of shared memory).
_SHM_KEEP_MAX = 48
_shm_keep = {}


def _keep_shared_mapping(t):
    if not t.is_shared():
        return
    key = t.untyped_storage().data_ptr()
    _shm_keep.pop(key, None)
    _shm_keep[key] = t                              # (insertion order = recency)
    while len(_shm_keep) > _SHM_KEEP_MAX:
        _shm_keep.pop(next(iter(_shm_keep)))


class HostPcmBatch:
    """One batch of decoded clips in HOST memory, as it comes out of ``DataLoader(dataset, num_workers>0)``: `clips`
    [B, S+1, n, ch] (mix last), `items` int64 [B] global item indices.  ``pin_memory()`` is what the loader's pin thread
    calls (``pin_memory=True``).  In the process that owns the GPU:
      * ``to_device()`` -> PcmBatch: the upload (asynchronous when the clips are page-locked) and the items' augmentation
        draws (data/dataset.py:198-199; keyed by seed, item and how often the item has been read -- the counters of the
        dataset object this process holds, so the draws are those of the ``num_workers=0`` loader in the same order);
      * unpacking (``train_features, gt_features = batch``, model_trainer.py:31-33 / training_ignite.ipynb cell 12) or
        ``features()`` runs the front-end on the uploaded batch: ONE launch for its B*(S+1) tracks."""
    __slots__ = ('clips', 'items', 'token', 'aug_seed', 'normalize', 'device', 'n_fft', 'hop', 'release', '_dev')

    def __init__(self, clips, items, token, aug_seed, normalize, device, n_fft=2048, hop=1024, release=None):
        self.clips, self.items, self.token, self.aug_seed, self.normalize, self.device = clips, items, token, aug_seed, normalize, device
        # release: the shared int32 word of a worker's re-usable block (MultitrackAudioDataset._worker_block), or None
        self.n_fft, self.hop, self.release, self._dev = n_fft, hop, release, None

    def __getstate__(self):
        return (self.clips, self.items, self.token, self.aug_seed, self.normalize, self.device, self.n_fft, self.hop, self.release)

    def __setstate__(self, state):
        (self.clips, self.items, self.token, self.aug_seed, self.normalize, self.device, self.n_fft, self.hop, self.release) = state
        self._dev = None

    def pin_memory(self):
        """Called by the DataLoader's pin-memory thread (``pin_memory=True``).  Page-locked memory from torch's caching
        host allocator, filled by a few plain copy threads -- NOT ``Tensor.pin_memory()``: its copy fans out over every
        OpenMP thread torch was given (128 on a box whose cgroup grants a 16-core share), the spinning team burns the
        process's CPU quota and the kernel throttles the whole process for the rest of each 100 ms period (measured here:
        115 ms per training step instead of 4.4; the same trap as staging._host_copy's)."""
        if self.clips.is_pinned():
            return self
        src = self.clips.contiguous()
        _keep_shared_mapping(src)
        dst = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        staging._host_copy(dst.view(-1).view(torch.uint8), src.view(-1).view(torch.uint8))
        if self.release is not None:
            self.release[0] = 0                # the worker may decode its next batch into this block
        return HostPcmBatch(dst, self.items, self.token, self.aug_seed, self.normalize, self.device, self.n_fft, self.hop)

    def draw_gains(self, device):
        """[B, S+1] augmentation gains of this batch's items on `device` (None without augmentation).  Advances the read
        counters of the live dataset: call once per batch."""
        if self.aug_seed is None:
            return None
        ds = _live_datasets.get(self.token)
        items = [int(i) for i in self.items]
        keys = ds._aug_keys(items) if ds is not None else items
        return features.augment_gains(self.aug_seed, self.clips.shape[1], items=keys, device=device)

    def to_device(self, device=None, out=None):
        """The batch in HBM as a PcmBatch (front-end not run yet).  out: a device buffer [>= B, S+1, n, ch] of the clips'
        dtype to upload into (ModelTrainer's staging slots); the copy is enqueued on the current stream."""
        if in_loader_worker():
            raise RuntimeError('HostPcmBatch.to_device() in a DataLoader worker / forked child: the GPU belongs to the '
                               'parent process -- hand the batch to the main process (that is what the loader does)')
        if self._dev is None:
            dev = torch.device(device if device is not None else self.device)
            B = self.clips.shape[0]
            if out is None:
                out = torch.empty(self.clips.shape, dtype=self.clips.dtype, device=dev)
            dst = out[:B]
            dst.copy_(self.clips, non_blocking=True)
            if self.release is not None and not self.clips.is_pinned():
                # (a pageable source has been read out when copy_ returns: the worker may re-use its block.  A consumer that
                # wants the host samples after this call keeps a copy -- `clips` of an un-pinned batch is the worker's block)
                self.release[0] = 0
            self._dev = PcmBatch(dst, self.draw_gains(dst.device), self.normalize, self.n_fft, self.hop)
        return self._dev

    def features(self, device=None):
        return self.to_device(device).features()

    def __iter__(self):
        return iter(self.features())


# DataLoader's default collate looks an item's type up here before its generic rules (a documented extension point):
# importing this module -- which unpickling / forking the dataset in a worker implies -- is what registers the two types
from torch.utils.data._utils.collate import default_collate_fn_map as _collate_map     # noqa: E402

_collate_map[PcmItem] = collate_pcm_items
_collate_map[DeviceItem] = lambda batch, *, collate_fn_map=None: DeviceBatch(torch.stack([b[0] for b in batch]),
                                                                             torch.stack([b[1] for b in batch]))


class _BatchLoader:
    def __init__(self, dataset, batch_size, indices, workers, drop_last, pcm=False):
        self.dataset, self.batch_size, self.indices, self.workers, self.drop_last = dataset, batch_size, indices, workers, drop_last
        self.pcm = pcm

    def __len__(self):
        n = len(self.dataset) if self.indices is None else len(self.indices)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        return self.dataset.iter_batches(self.batch_size, self.indices, self.workers, self.drop_last, pcm=self.pcm)

"""Host mirror of the reference's ordered chunk replay, MapPoolOrd (src/map_pool_ord.cpp:61-112): every channel has a
queue of reads; each update() hands the next chunk of the front read of every channel to the pool, and a read
leaves its queue when the pool reports it finished.  Deterministic by construction (a chunk is added only after the
previous one is mapped), which is what makes bit-exact comparison with the reference possible."""
import numpy as np

from uncalled_amd import capi


class MapPoolOrd:
    def __init__(self, index, n_channels=512, params=None, chunk_len=None):
        self.rt = capi.Realtime(index, n_channels, params)
        p = self.rt.params
        self.chunk_len = chunk_len or int(p.chunk_time * p.sample_rate)     # ReadBuffer::PRMS.chunk_len()
        self.queues = [[] for _ in range(n_channels)]                       # channels_[ch]: (number, raw, calib, key)
        self.chunk_idx = [0] * n_channels
        self.chunks_used = {}                                                # key -> chunks the read was given before it was decided

    def add_read(self, channel, number, raw_i16, calib, key=None):
        """calib = (range, offset, digitisation) of the channel; key identifies the read in the results."""
        self.queues[channel].append((number, np.ascontiguousarray(raw_i16, dtype=np.int16), calib, key))

    def running(self):
        return any(self.queues)

    def update(self):
        """One round over all channels; returns [(key, RT_RESULT record)] for reads that finished in this round."""
        chunks, parts, owners, off = [], [], [], 0
        for ch, q in enumerate(self.queues):
            if not q:
                continue
            number, raw, cal, key = q[0]
            ci = self.chunk_idx[ch]
            st = min(ci * self.chunk_len, raw.size)                          # ReadBuffer::get_chunk
            ln = min(self.chunk_len, raw.size - st)
            flags = capi.RT_FIRST if ci == 0 else 0
            if st + ln >= raw.size:
                flags |= capi.RT_LAST                                         # the next get_chunk would be empty
            c = np.zeros(1, dtype=capi.RT_CHUNK)[0]
            c["channel"], c["read_number"], c["flags"], c["n_samples"], c["offset"] = ch, number, flags, ln, off
            c["calib"]["range"], c["calib"]["offset"], c["calib"]["digitisation"] = cal
            chunks.append(c)
            parts.append(raw[st:st + ln])
            owners.append(ch)
            off += ln
        if not chunks:
            return []
        res = self.rt.process_chunks(np.array(chunks, dtype=capi.RT_CHUNK), np.concatenate(parts) if off else np.zeros(1, np.int16))
        out = []
        for ch, r in zip(owners, res):
            if r["state"] == capi.RT_MAPPING:
                self.chunk_idx[ch] += 1
            else:
                self.chunks_used[self.queues[ch][0][3]] = self.chunk_idx[ch] + 1
                out.append((self.queues[ch][0][3], r.copy()))
                self.queues[ch].pop(0)
                self.chunk_idx[ch] = 0
        return out

//! Batched queries: what `index.search(p).count()` and `.iter_matches().map(|m| m.locate())`
//! (`benches/count.rs:33`, `benches/locate.rs:42-46`) become when there are many patterns.
use super::backend::{last_error, GpuBackend, GpuCharacter};
use super::ffi;

/// `(s, e)` and `count = e - s` of every pattern (`wrapper.rs:126-134`)
pub struct BatchCounts {
    pub s: Vec<u64>,
    pub e: Vec<u64>,
    pub counts: Vec<u64>,
}

/// `positions[offsets[k]..offsets[k + 1]]` = pattern k's `locate()` values in the reference's
/// iteration order (`wrapper.rs:203-217`: suffix-array order)
pub struct BatchPositions {
    pub offsets: Vec<u64>,
    pub positions: Vec<u64>,
}

impl<C: GpuCharacter> GpuBackend<C> {
    /// `patterns.iter().map(|p| index.search(p))` in one call
    pub fn search_many<P: AsRef<[C]>>(&self, patterns: &[P]) -> BatchCounts {
        let mut flat: Vec<C> = Vec::new();
        let mut off = Vec::with_capacity(patterns.len() + 1);
        off.push(0u64);
        for p in patterns {
            flat.extend_from_slice(p.as_ref());
            off.push(flat.len() as u64);
        }
        let n = patterns.len();
        let (mut s, mut e, mut counts) = (vec![0u64; n], vec![0u64; n], vec![0u64; n]);
        let rc = unsafe {
            ffi::fmx_count_batch(
                self.h,
                flat.as_ptr() as *const _,
                off.as_ptr(),
                n as u64,
                std::ptr::null(),
                s.as_mut_ptr(),
                e.as_mut_ptr(),
                counts.as_mut_ptr(),
            )
        };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        BatchCounts { s, e, counts }
    }

    /// every match position of every pattern of a `search_many` result
    pub fn locate_many(&self, found: &BatchCounts) -> BatchPositions {
        let n = found.s.len();
        let mut offsets = Vec::with_capacity(n + 1);
        let mut acc = 0u64;
        offsets.push(0);
        for c in &found.counts {
            acc += c;
            offsets.push(acc);
        }
        let mut positions = vec![0u64; acc as usize];
        let rc = unsafe {
            ffi::fmx_locate_batch(
                self.h,
                found.s.as_ptr(),
                found.e.as_ptr(),
                n as u64,
                offsets.as_ptr(),
                positions.as_mut_ptr(),
            )
        };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        BatchPositions { offsets, positions }
    }
}

/// The index replicated over several GPUs of one node (BASELINE config 5): patterns are independent
/// (`wrapper.rs:103-124` reads only immutable index state), so a batch shards contiguously over the replicas --
/// pattern k of N on replica k * G / N -- and every shard's results land in place in the caller's vectors: the
/// same `BatchCounts` / `BatchPositions` as `search_many` / `locate_many` on one handle, bit for bit.
pub struct GpuReplicas<C> {
    replicas: Vec<GpuBackend<C>>,
}

impl<C: GpuCharacter> GpuBackend<C> {
    /// a second handle with its own copy of the HBM arrays on `device` (device-to-device copy, no rebuild)
    pub fn replicate(&self, device: i32) -> GpuBackend<C> {
        let mut h = std::ptr::null_mut();
        let rc = unsafe { ffi::fmx_replicate(self.h, device, &mut h) };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        GpuBackend::from_raw(h)
    }
}

impl<C: GpuCharacter> GpuReplicas<C> {
    /// `first` stays replica 0; one more replica per entry of `devices`
    pub fn new(first: GpuBackend<C>, devices: &[i32]) -> Self {
        let mut replicas = Vec::with_capacity(devices.len() + 1);
        let more: Vec<GpuBackend<C>> = devices.iter().map(|&d| first.replicate(d)).collect();
        replicas.push(first);
        replicas.extend(more);
        GpuReplicas { replicas }
    }

    pub fn len(&self) -> usize {
        self.replicas.len()
    }

    fn handles(&self) -> Vec<*mut ffi::FmxIndex> {
        self.replicas.iter().map(|r| r.h).collect()
    }

    /// `search_many` over all replicas (`fmx_count_batch_multi`)
    pub fn search_many_sharded<P: AsRef<[C]>>(&self, patterns: &[P]) -> BatchCounts {
        let mut flat: Vec<C> = Vec::new();
        let mut off = Vec::with_capacity(patterns.len() + 1);
        off.push(0u64);
        for p in patterns {
            flat.extend_from_slice(p.as_ref());
            off.push(flat.len() as u64);
        }
        let n = patterns.len();
        let (mut s, mut e, mut counts) = (vec![0u64; n], vec![0u64; n], vec![0u64; n]);
        let hs = self.handles();
        let rc = unsafe {
            ffi::fmx_count_batch_multi(
                hs.as_ptr(),
                hs.len() as u32,
                flat.as_ptr() as *const _,
                off.as_ptr(),
                n as u64,
                std::ptr::null(),
                s.as_mut_ptr(),
                e.as_mut_ptr(),
                counts.as_mut_ptr(),
            )
        };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        BatchCounts { s, e, counts }
    }

    /// `locate_many` over all replicas (`fmx_locate_batch_multi`)
    pub fn locate_many_sharded(&self, found: &BatchCounts) -> BatchPositions {
        let n = found.s.len();
        let mut offsets = Vec::with_capacity(n + 1);
        let mut acc = 0u64;
        offsets.push(0);
        for c in &found.counts {
            acc += c;
            offsets.push(acc);
        }
        let mut positions = vec![0u64; acc as usize];
        let hs = self.handles();
        let rc = unsafe {
            ffi::fmx_locate_batch_multi(
                hs.as_ptr(),
                hs.len() as u32,
                found.s.as_ptr(),
                found.e.as_ptr(),
                n as u64,
                offsets.as_ptr(),
                positions.as_mut_ptr(),
            )
        };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        BatchPositions { offsets, positions }
    }
}

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

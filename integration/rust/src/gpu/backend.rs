//! `GpuBackend`: the crate-private backend traits (`src/backend.rs:5-31`) over libfmx.
use std::ffi::CStr;
use std::marker::PhantomData;

use super::ffi;
use crate::backend::{HasMultiPieces, HasPosition, SearchIndexBackend};
use crate::character::Character;
use crate::error::Error;
use crate::heap_size::HeapSize;
use crate::piece::PieceId;
use crate::text::Text;

/// which of the reference's index types the handle stands for (`frontend.rs:110-193`)
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum GpuIndexKind {
    /// `FMIndex` / `FMIndexWithLocate` (`src/fm_index.rs`)
    Fm = ffi::FMX_KIND_FM as isize,
    /// `RLFMIndex` / `RLFMIndexWithLocate` (`src/rlfmi.rs`)
    Rlfm = ffi::FMX_KIND_RLFM as isize,
    /// `FMIndexMultiPieces` / `FMIndexMultiPiecesWithLocate` (`src/multi_pieces.rs`)
    Multi = ffi::FMX_KIND_MULTI as isize,
}

/// Symbol widths libfmx takes natively (`Character`, `character.rs:38-42`).
pub trait GpuCharacter: Character {
    const SYM_BYTES: u32;
    fn from_u64_lossy(v: u64) -> Self;
}
impl GpuCharacter for u8 {
    const SYM_BYTES: u32 = 1;
    fn from_u64_lossy(v: u64) -> u8 { v as u8 }
}
impl GpuCharacter for u16 {
    const SYM_BYTES: u32 = 2;
    fn from_u64_lossy(v: u64) -> u16 { v as u16 }
}
impl GpuCharacter for u32 {
    const SYM_BYTES: u32 = 4;
    fn from_u64_lossy(v: u64) -> u32 { v as u32 }
}
impl GpuCharacter for u64 {
    const SYM_BYTES: u32 = 8;
    fn from_u64_lossy(v: u64) -> u64 { v }
}

pub struct GpuBackend<C> {
    pub(super) h: *mut ffi::FmxIndex,
    _c: PhantomData<C>,
}
// the handle is immutable after build; any number of threads may query it (include/fmx.h)
unsafe impl<C> Send for GpuBackend<C> {}
unsafe impl<C> Sync for GpuBackend<C> {}

pub(super) fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::fmx_last_error()) }.to_string_lossy().into_owned()
}

impl<C: GpuCharacter> GpuBackend<C> {
    /// replaces `FMIndexBackend::new` / `RLFMIndexBackend::new` (`fm_index.rs:25-42`,
    /// `rlfmi.rs:30-96`); `level = None` for the count-only types.
    pub fn new<T: AsRef<[C]>>(
        text: &Text<C, T>,
        kind: GpuIndexKind,
        level: Option<usize>,
        device: i32,
    ) -> Result<Self, Error> {
        let mut h = std::ptr::null_mut();
        let t = text.text();
        let rc = unsafe {
            ffi::fmx_build(
                t.as_ptr() as *const _,
                t.len() as u64,
                C::SYM_BYTES,
                text.max_character().into_u64(),
                kind as u32,
                level.map(|l| l as u32).unwrap_or(ffi::FMX_NO_LOCATE),
                0,
                device,
                &mut h,
            )
        };
        match rc {
            ffi::FMX_OK => Ok(GpuBackend { h, _c: PhantomData }),
            // the reference's two messages (sais.rs:128-139)
            ffi::FMX_ERR_TEXT_START_ZERO => {
                Err(Error::InvalidText("the given text must not start with zero character"))
            }
            ffi::FMX_ERR_TEXT_END_ZERO => {
                Err(Error::InvalidText("the given text must end with exactly one zero character"))
            }
            // where the reference panics (symbol > max_character: sais.rs:18) or cannot happen
            _ => panic!("libfmx: {}", last_error()),
        }
    }

    /// takes ownership of a handle libfmx made (`fmx_replicate`, `fmx_load`)
    pub(super) fn from_raw(h: *mut ffi::FmxIndex) -> Self {
        GpuBackend { h, _c: PhantomData }
    }

    pub fn level(&self) -> Option<usize> {
        match unsafe { ffi::fmx_level(self.h) } {
            ffi::FMX_NO_LOCATE => None,
            l => Some(l as usize),
        }
    }
}

impl<C> Drop for GpuBackend<C> {
    fn drop(&mut self) {
        unsafe { ffi::fmx_free(self.h) }
    }
}

/// a failing one-element call returns u64::MAX and sets fmx_last_error(); the reference's methods
/// are infallible by signature and panic on a bad argument (`fm_index.rs:94`: `cs[c]`)
#[inline]
fn checked(v: u64) -> u64 {
    if v == u64::MAX {
        panic!("libfmx: {}", last_error());
    }
    v
}

impl<C: GpuCharacter> SearchIndexBackend for GpuBackend<C> {
    type C = C;

    fn get_l(&self, i: usize) -> C {
        C::from_u64_lossy(checked(unsafe { ffi::fmx_get_l(self.h, i as u64) }))
    }
    fn lf_map(&self, i: usize) -> usize {
        checked(unsafe { ffi::fmx_lf_map(self.h, i as u64) }) as usize
    }
    fn lf_map2(&self, c: C, i: usize) -> usize {
        checked(unsafe { ffi::fmx_lf_map2(self.h, c.into_u64(), i as u64) }) as usize
    }
    fn get_f(&self, i: usize) -> C {
        C::from_u64_lossy(checked(unsafe { ffi::fmx_get_f(self.h, i as u64) }))
    }
    fn fl_map(&self, i: usize) -> Option<usize> {
        // Some for FM / RLFM (fm_index.rs:114-120, rlfmi.rs:160-169); None on a multi-pieces index when
        // F[i] is the piece separator (multi_pieces.rs:176-187).  The batch form tells the two apart: its
        // return code is the error, the all-ones value is None
        let (row, mut out) = (i as u64, u64::MAX);
        let rc = unsafe { ffi::fmx_fl_map_batch(self.h, &row, 1, &mut out) };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        if out == u64::MAX {
            None
        } else {
            Some(out as usize)
        }
    }
    fn len(&self) -> usize {
        unsafe { ffi::fmx_len(self.h) as usize }
    }

    /// the loop of `SearchWrapper::search` (`wrapper.rs:103-124`), moved into the backend as the
    /// crate's own TODO asks (`wrapper.rs:104`): ONE launch for the whole pattern instead of
    /// 2 x len one-element launches.  See gpu-backend.patch for the provided method this overrides.
    fn search_range(&self, pattern: &[C], s: usize, e: usize) -> (usize, usize) {
        let off = [0u64, pattern.len() as u64];
        let se = [s as u64, e as u64];
        let (mut os, mut oe) = (0u64, 0u64);
        let rc = unsafe {
            ffi::fmx_count_batch(
                self.h,
                pattern.as_ptr() as *const _,
                off.as_ptr(),
                1,
                se.as_ptr(),
                &mut os,
                &mut oe,
                std::ptr::null_mut(),
            )
        };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        (os as usize, oe as usize)
    }

    /// the rows `iter_matches` visits (`wrapper.rs:203-217`): `s..e`, or for a prefix-only search
    /// (`search_prefix` / `search_exact`, `wrapper.rs:57-82`) the rows of the range whose L symbol is 0 --
    /// counted and listed on the device (`fmx_match_counts` + `fmx_match_rows`) instead of one `get_l`
    /// call per row.  See gpu-backend.patch for the provided method this overrides.
    fn match_rows<'a>(
        &'a self,
        s: usize,
        e: usize,
        match_prefix_only: bool,
    ) -> Box<dyn Iterator<Item = usize> + 'a> {
        if !match_prefix_only {
            return Box::new(s..e);
        }
        let (ss, ee) = (s as u64, e as u64);
        let mut count = 0u64;
        let rc = unsafe { ffi::fmx_match_counts(self.h, &ss, &ee, 1, 1, &mut count) };
        if rc != ffi::FMX_OK {
            panic!("libfmx: {}", last_error());
        }
        let mut rows = vec![0u64; count as usize];
        if count > 0 {
            let off = [0u64, count];
            let rc = unsafe {
                ffi::fmx_match_rows(self.h, &ss, &ee, 1, 1, off.as_ptr(), rows.as_mut_ptr())
            };
            if rc != ffi::FMX_OK {
                panic!("libfmx: {}", last_error());
            }
        }
        Box::new(rows.into_iter().map(|r| r as usize))
    }
}

/// `FMIndexMultiPieces*` (`multi_pieces.rs:201-224`): built with `GpuIndexKind::Multi`
impl<C: GpuCharacter> HasMultiPieces for GpuBackend<C> {
    fn piece_id(&self, i: usize) -> PieceId {
        PieceId::from(checked(unsafe { ffi::fmx_piece_id(self.h, i as u64) }) as usize)
    }
    fn pieces_count(&self) -> usize {
        unsafe { ffi::fmx_pieces_count(self.h) as usize }
    }
}

impl<C: GpuCharacter> HasPosition for GpuBackend<C> {
    fn get_sa(&self, i: usize) -> usize {
        checked(unsafe { ffi::fmx_get_sa(self.h, i as u64) }) as usize
    }
}

impl<C> HeapSize for GpuBackend<C> {
    fn heap_size(&self) -> usize {
        unsafe { ffi::fmx_index_bytes(self.h) as usize }
    }
}

/// `Match::iter_chars_backward` (`wrapper.rs:154-161`) refilled 64 characters per launch
pub struct GpuBackwardChars<'a> {
    backend: &'a GpuBackend<u8>,
    row: u64,
    buf: [u8; 64],
    at: usize,
}
impl<'a> GpuBackwardChars<'a> {
    pub fn new(backend: &'a GpuBackend<u8>, row: usize) -> Self {
        GpuBackwardChars { backend, row: row as u64, buf: [0; 64], at: 64 }
    }
}
impl Iterator for GpuBackwardChars<'_> {
    type Item = u8;
    fn next(&mut self) -> Option<u8> {
        if self.at == 64 {
            let (mut len, mut next) = (0u64, 0u64);
            let rc = unsafe {
                ffi::fmx_extract_batch(
                    self.backend.h,
                    &self.row,
                    1,
                    64,
                    0,
                    self.buf.as_mut_ptr() as *mut _,
                    &mut len,
                    &mut next,
                )
            };
            if rc != ffi::FMX_OK {
                panic!("libfmx: {}", last_error());
            }
            self.row = next;
            self.at = 0;
        }
        self.at += 1;
        Some(self.buf[self.at - 1])
    }
}

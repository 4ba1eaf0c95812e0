//! MI355X backend for the count / locate path (feature `gpu`).
//!
//! * [`ffi`]     -- raw binding of `include/fmx.h`, generated from the header.
//! * [`backend`] -- `GpuBackend`: `impl SearchIndexBackend + HasPosition + HasMultiPieces + HeapSize`
//!                  (`src/backend.rs:5-40`), so `SearchIndexWrapper<GpuBackend<_>>` -- and with it the
//!                  public `Search` / `Match` / `MatchWithLocate` / `MatchWithPieceId` traits and the
//!                  multi-pieces searches (`search_prefix` / `search_suffix` / `search_exact`) -- run unchanged.
//! * [`batch`]   -- what the GPU is for: many patterns per call (`search_many`, `locate_many`), and one batch over
//!                  the replicas of an index on several GPUs of a node (`GpuReplicas::search_many_sharded`,
//!                  `locate_many_sharded`: BASELINE config 5 from one Rust caller).
pub mod backend;
pub mod batch;
pub mod ffi;

pub use backend::{GpuBackend, GpuIndexKind};
pub use batch::{BatchCounts, BatchPositions, GpuReplicas};

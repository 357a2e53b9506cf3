//! Binding of include/zkmi355_rccl.h (libzkmi355_rccl.so): the collective of one proof over several GPUs on RCCL / xGMI — `mod rccl;` beside `mod mi355x;`.
//! `zk_rccl_allgather` IS a `ZkAllgatherFn` (pk_desc.rs): it goes into `ZkPlonkPkHost::allgather` with the rank's communicator as `allgather_user`.
//! Uncompiled in the build image (no rustc there); tests/test_shim_abi.py diffs every declaration below against the header.
use std::ffi::{c_void, CStr};
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct ZkRcclComm {
    _p: [u8; 0],
}

pub const ZK_RCCL_UNIQUE_ID_BYTES: usize = 128;

#[link(name = "zkmi355_rccl")]
extern "C" {
    pub fn zk_rccl_unique_id(out: *mut c_void) -> c_int;
    pub fn zk_rccl_comm_create(world: u32, rank: u32, unique_id: *const c_void, device: c_int, timeout_ms: u32, comm: *mut *mut ZkRcclComm) -> c_int;
    pub fn zk_rccl_comm_init_all(ndev: u32, devices: *const c_int, timeout_ms: u32, comms: *mut *mut ZkRcclComm) -> c_int;
    pub fn zk_rccl_allgather(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> c_int;
    pub fn zk_rccl_comm_world(comm: *const ZkRcclComm) -> u32;
    pub fn zk_rccl_comm_rank(comm: *const ZkRcclComm) -> u32;
    pub fn zk_rccl_comm_calls(comm: *const ZkRcclComm) -> u64;
    pub fn zk_rccl_last_error(comm: *const ZkRcclComm) -> *const c_char;
    pub fn zk_rccl_comm_destroy(comm: *mut ZkRcclComm);
}

/// `zk_rccl_allgather` behind the safe function-pointer type of the descriptor (a foreign function is `unsafe extern "C" fn`)
extern "C" fn gather(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> c_int {
    unsafe { zk_rccl_allgather(user, send_dev, recv_dev, bytes) }
}

/// One process, `ndev` GPUs (ncclCommInitAll): communicator r belongs to the proving thread of device r.
pub struct Clique(pub Vec<*mut ZkRcclComm>);
unsafe impl Send for Clique {}
unsafe impl Sync for Clique {}

impl Clique {
    pub fn init_all(ndev: u32, timeout_ms: u32) -> Result<Clique, String> {
        let mut comms = vec![std::ptr::null_mut(); ndev as usize];
        let rc = unsafe { zk_rccl_comm_init_all(ndev, std::ptr::null(), timeout_ms, comms.as_mut_ptr()) };
        if rc != 0 {
            return Err(unsafe { CStr::from_ptr(zk_rccl_last_error(std::ptr::null())) }.to_string_lossy().into_owned());
        }
        Ok(Clique(comms))
    }
    /// what goes into ZkPlonkPkHost { allgather, allgather_user } of rank r
    pub fn hook(&self, r: usize) -> (crate::pk_desc::ZkAllgatherFn, *mut c_void) {
        (gather, self.0[r] as *mut c_void)
    }
}

impl Drop for Clique {
    fn drop(&mut self) {
        for c in self.0.drain(..) {
            unsafe { zk_rccl_comm_destroy(c) };
        }
    }
}

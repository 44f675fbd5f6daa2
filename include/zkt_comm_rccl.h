/* zkt_comm_rccl.h -- the RCCL transport of a proof sharded across the GPUs of one node, as an OPTIONAL second library
 * (libzkt_comm_rccl.so).  libzkt_plonk_hip.so itself links no transport: it talks to a zkt_comm_vtable
 * (zkt_plonk.h, "one proof across the GPUs of a node").  A host that has no communicator of its own -- the reference's
 * caller is a Rust binary, plonk-core/src/plonk.rs:94-111 under bin/src/main.rs:285-293 -- links this library and gets a
 * vtable whose all_gather is ncclAllGather over xGMI on the proving context's stream.
 *
 * One process per GPU.  Rank 0 calls zkt_comm_rccl_unique_id and ships the 128 bytes to the other ranks by any means it
 * has (a file, a socket, an environment variable of the launcher); every rank then calls zkt_comm_rccl_create with the
 * same id (collective: it returns when all `world` ranks have joined), zkt_comm_rccl_vtable, and hands the vtable to
 * zkt_ctx_set_comm.  The vtable stays valid until zkt_comm_rccl_destroy; detach it (zkt_ctx_set_comm(ctx, NULL)) first.
 *
 * STATUS: with world = 1 this path runs in the GPU tests (the sharded prover's device branch, capi.hip
 * comm_all_gather_dev -> ncclAllGather).  With world > 1 it has NEVER RUN ON HARDWARE: the development pool hands out
 * one GPU per machine.  The exchange is the same two-line call either way, but treat it as unverified.
 *
 * A process must hold ONE copy of RCCL and ONE HIP runtime.  A host that also loads PyTorch (whose wheel ships its own
 * librccl.so.1 / libamdhip64.so.7) must let that copy be the one that is mapped first; a plain host gets /opt/rocm's. */
#ifndef ZKT_COMM_RCCL_H
#define ZKT_COMM_RCCL_H

#include <stddef.h>
#include <stdint.h>

#include "zkt_plonk.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ZKT_COMM_RCCL_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct zkt_comm_rccl zkt_comm_rccl;

/* ncclGetUniqueId: call on ONE rank, distribute the bytes. */
int zkt_comm_rccl_unique_id(uint8_t out[ZKT_COMM_RCCL_UNIQUE_ID_BYTES]);
/* hipSetDevice(device) + ncclCommInitRank; collective over the `world` ranks (1, 2, 4 or 8).  Returns ZKT_OK or
 * ZKT_ERR_COMM / ZKT_ERR_HIP / ZKT_ERR_INVALID_ARGUMENT; *out is NULL on failure. */
int zkt_comm_rccl_create(const uint8_t id[ZKT_COMM_RCCL_UNIQUE_ID_BYTES], int rank, int world, int device, zkt_comm_rccl** out);
/* The communicator as the library wants it: device_buffers = 1; all_gather(send, recv, bytes, on_device, stream):
 *   on_device = 1: ncclAllGather of `bytes` chars per rank on `stream`, then the stream is synchronised (the vtable's
 *                  contract: complete on return).  send == recv + rank * bytes is RCCL's in-place form and is passed
 *                  through; any other overlap of send with recv is staged through a private device buffer.
 *   on_device = 0: host buffers, staged through a private device buffer on a private stream.
 * all_gather_async(d_send, d_recv, bytes, stream): the device form without the synchronisation: enqueued on `stream` and
 * left there (zkt_comm_vtable's optional stream-ordered entry; the library then pipelines the quotient exchange).
 * A local HIP / RCCL failure inside an exchange aborts the communicator (ncclCommAbort) so that the peers' collectives
 * fail instead of waiting for this rank; every later call on it fails.  One host thread at a time per communicator. */
int zkt_comm_rccl_vtable(zkt_comm_rccl* comm, zkt_comm_vtable* out);
/* ncclCommDestroy + release of the staging buffers. */
void zkt_comm_rccl_destroy(zkt_comm_rccl* comm);
/* Text of the last RCCL / HIP failure on this communicator ("" if none); owned by the communicator. */
const char* zkt_comm_rccl_last_error(const zkt_comm_rccl* comm);

#ifdef __cplusplus
}
#endif
#endif

// Shape of one locality block's LDS working set (shared by the host-side blocking and the blocked kernels).
#pragma once
#ifndef MMW_BLK_THREADS
#define MMW_BLK_THREADS 1024   // workgroup size of the blocked kernels
#endif
#ifndef MMW_BLK_UNION
#define MMW_BLK_UNION 416      // staged dense rows per block (416 x 128 B x 3 workgroups fit the 160 KiB LDS)
#endif
#ifndef MMW_BLK_META
#define MMW_BLK_META 38144     // LDS bytes for the block's staged (row offset, value) entries
#endif

// Satellites: LDS state (member `sat` of Smem in the satellite builds, NSAT > 0).  See smj_sat.h.
#pragma once
#if NSAT > 0
#define NXV (NVS + 6 * NXS)   // order of the dense Newton system of a step in which satellites are coupled: the main columns + NXS satellites
#define NIT 20                // row items (a contact's rows / a friction-loss row / a limit row) one satellite can hold in a step
#define NSS 8                 // satellite-satellite contacts of a step
// satellite 6-vectors (one per satellite and field)
enum { SX_V = 0, SX_QA, SX_MA, SX_GRAD, SX_SRCH, SX_MV, SX_G, SX_TMP, SX_N };
enum { ITEM_N = 7, ITEM_SLOT = 8, ITEM_CONTACT = 16, ITEM_MAIN = 32 /* the other body of the contact belongs to the main tree */ };
struct SatMem {
  float x[SX_N][NSAT][6];   // qvel | qacc (qacc_warmstart at the start of the solve) | M qacc | gradient | search | M search | qfrc_smooth | scratch: J'f, right-hand sides
  float q[NSAT][7];
  float Mb[NSAT][21], Hb[NSAT][21];   // mass block and Newton block of the satellite (packed lower triangle, row major); 1-dof satellites: identity padded
  float wax[NSAT][3], wanc[NSAT][3];  // world joint axis / anchor of a hinge or slide satellite
  int jtype[NSAT], ndof[NSAT], body[NSAT];
  int ext[NSAT];            // slot in the dense extension of this step's Newton system (-1: the block is solved on the satellite's own lane)
  int xs[NXS];              // satellites of the extension, slot order
  int nitem[NSAT];
  unsigned char irow[NSAT][NIT], iinf[NSAT][NIT], icon[NSAT][NIT];   // first row | rows (ITEM_N), slot, flags | contact index
  int sscon[NSS];           // contacts between two satellites
  float Js[NEFC][2][6];     // the satellite columns of a constraint row: slot u = satellite esat[row][u]
  signed char esat[NEFC][2];
  short erec[NEFC];         // row record (DevModel::k_rowrec) of a static / limit row -- with satellite rows moved behind the dense rows a row no longer sits at its record's index
};
#endif
